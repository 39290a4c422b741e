'use strict';
// End-to-end request rate of GET /api/calendar through the Node host on one Node thread (/root/reference/server/index.js:293-302
// is the route kept).  Two workloads:
//   node bench_requests.js [users] [sessionsPerUser] [requests]      a store built session by session (every row has a token)
//   node bench_requests.js --cfg3 [rows] [users] [requests] [concurrency]
//        BASELINE config 3's table (10^8 sessions / 10^5 users, generated on the device as the store's base) + one real session per
//        requesting user; `concurrency` requests in flight over keep-alive sockets, request coalescing ON: the requests of one
//        event-loop turn become ONE batched device scan (every request its own clock: up to 64 distinct queries per table pass),
//        one native fetch for all their rows, one native serialiser call each.
// Prints one JSON line with requests/s and the split of a request's time: device scan / fetch / serialiser / everything else (HTTP,
// cookie + sha256, role check, sockets).
process.env.TZ = 'UTC';
const http = require('http');
const {createStore} = require('./sessionStore');
const {createFeedService} = require('./feedService');
const {createServer} = require('./server');
const dc = require('./disciplineConfig');

const realNow = Date.now;
let fakeNow = 1750000000000;
Date.now = () => fakeNow;

async function classic(){
  const U = Number(process.argv[2] || 2000), S = Number(process.argv[3] || 50), R = Number(process.argv[4] || 2000);
  const store = createStore();
  const users = new Map();
  const cookies = [];
  const discs = ['drones', 'audio', 'video', 'lighting'];
  for(let s = 0; s < S; s++){
    for(let u = 0; u < U; u++){
      const id = 'user-' + u;
      if(s === 0){ users.set(id, {id, roles: ['drones.lead']}); }
      fakeNow += 1;
      const made = store.createSession(id, discs[(u + s) % discs.length]);
      if(s === S - 1){ cookies.push(store.SESSION_COOKIE_NAME + '=' + made.token); }
    }
  }
  const feeds = createFeedService(store);
  const server = createServer({store, feeds, findUserById: id => users.get(id) || null});
  await new Promise(r => server.listen(0, '127.0.0.1', r));
  const port = server.address().port;
  const agent = new http.Agent({keepAlive: true, maxSockets: 1});
  const get = cookie => new Promise((resolve, reject) => {
    http.get({host: '127.0.0.1', port, path: '/api/calendar', agent, headers: {cookie}}, res => {
      let n = 0; res.on('data', c => { n += c.length; }); res.on('end', () => resolve({status: res.statusCode, bytes: n}));
    }).on('error', reject);
  });
  fakeNow += 1000;
  const first = await get(cookies[0]);
  if(first.status !== 200){ throw new Error('unexpected status ' + first.status); }
  const t0 = process.hrtime.bigint();
  let bytes = 0;
  for(let r = 0; r < R; r++){
    fakeNow += 1;                                  // every request at its own instant: one device scan each
    const got = await get(cookies[r % cookies.length]);
    bytes += got.bytes;
  }
  const dt = Number(process.hrtime.bigint() - t0) / 1e9;
  const scans = feeds.scansRun();
  // the same requests at ONE instant: they share a scan
  const t1 = process.hrtime.bigint();
  for(let r = 0; r < R; r++){ await get(cookies[r % cookies.length]); }
  const dt2 = Number(process.hrtime.bigint() - t1) / 1e9;
  console.log(JSON.stringify({sessions: U * S, users: U, requests: R, node: process.version,
    requests_per_sec_one_scan_each: R / dt, ms_per_request_one_scan_each: dt * 1e3 / R, device_scans: scans, body_bytes_per_request: bytes / R,
    requests_per_sec_shared_scan: R / dt2, ms_per_request_shared_scan: dt2 * 1e3 / R, device_scans_shared_phase: feeds.scansRun() - scans}));
  server.close(); agent.destroy(); store.close();
}

async function cfg3(){
  const N = Number(process.argv[3] || 1e8), U = Number(process.argv[4] || 1e5), R = Number(process.argv[5] || 20000), C = Number(process.argv[6] || 64);
  const D = dc.DISCIPLINES.length;                   // the host's discipline table (7): the serialiser's per-discipline constants
  const T0 = 1700000000000, DAY = 86400000;
  const now = T0 - 6 * 3600000, cutoff = T0 - 61 * DAY;    // the spec query of SURVEY.md 8d
  fakeNow = now;
  const t_load = process.hrtime.bigint();
  const store = createStore({base: {rows: N, users: U, disc: D}});
  const nClients = Math.min(2000, U);
  const users = new Map();
  const cookies = [];
  for(let k = 0; k < nClients; k++){
    const id = 'user-' + Math.floor(k * (U / nClients));
    users.set(id, {id, roles: ['drones.crew']});
    cookies.push(store.SESSION_COOKIE_NAME + '=' + store.createSession(id, dc.DISCIPLINES[k % D].id).token);
  }
  const feeds = createFeedService(store);
  let served = 0;
  // every request samples its own clock (sessionStore.js:67 does, once per scan): requests of one turn differ by up to a minute
  const server = createServer({store, feeds, coalesce: true, findUserById: id => users.get(id) || null,
    query: () => ({now: now - (served++ % 64) * 977, cutoff})});
  await new Promise(r => server.listen(0, '127.0.0.1', r));
  const load_s = Number(process.hrtime.bigint() - t_load) / 1e9;
  const port = server.address().port;
  const agent = new http.Agent({keepAlive: true, maxSockets: C});
  const get = cookie => new Promise((resolve, reject) => {
    http.get({host: '127.0.0.1', port, path: '/api/calendar', agent, headers: {cookie}}, res => {
      let n = 0; res.on('data', c => { n += c.length; }); res.on('end', () => resolve({status: res.statusCode, bytes: n}));
    }).on('error', reject);
  });
  const run = async (count) => {
    let next = 0, bytes = 0, bad = 0;
    const worker = async () => {
      while(next < count){
        const r = next++;
        const got = await get(cookies[r % cookies.length]);
        bytes += got.bytes;
        if(got.status !== 200){ bad++; }
      }
    };
    const ws = [];
    for(let k = 0; k < C; k++){ ws.push(worker()); }
    await Promise.all(ws);
    return {bytes, bad};
  };
  await run(Math.min(2000, R));                      // warm-up (sockets, the device's adaptive choices)
  const b0 = feeds.batchesRun(), tm0 = feeds.timing();
  const cpu0 = process.cpuUsage();
  const t0 = process.hrtime.bigint();
  const got = await run(R);
  const dt = Number(process.hrtime.bigint() - t0) / 1e9;
  const cpu = process.cpuUsage(cpu0);
  const tm = feeds.timing(), batches = feeds.batchesRun() - b0;
  const per = ns => ns / 1e3 / R;                    // microseconds per request
  const scan_us = per(tm.scanNs - tm0.scanNs), fetch_us = per(tm.fetchNs - tm0.fetchNs), ser_us = per(tm.serializeNs - tm0.serializeNs);
  console.log(JSON.stringify({workload: 'GET /api/calendar, BASELINE config 3 table as the store\'s base (' + N + ' sessions / ' + U + ' users, ' + D +
      ' disciplines: the host table), spec query, every request its own clock', node: process.version, requests: R, concurrency: C,
    distinct_clients: nClients, requests_per_sec: R / dt, feeds_per_sec: R / dt, us_per_request: dt * 1e6 / R, non_200: got.bad,
    body_bytes_per_request: got.bytes / R, batches, requests_per_batch: R / Math.max(batches, 1),
    split_us_per_request: {device_batched_scan: scan_us, device_fetch_of_all_rows: fetch_us, native_serialiser: ser_us,
      http_auth_sockets_and_the_rest: dt * 1e6 / R - scan_us - fetch_us - ser_us},
    node_cpu_us_per_request: (cpu.user + cpu.system) / R, table_ready_seconds: load_s,
    note: 'one Node thread drives client and server sides of the loopback sockets; the device answers a batch of up to 64 distinct queries ' +
      'with one table pass + one fetch, so its share shrinks with the batch size — the request rate is the Node HTTP stack\'s'}));
  server.close(); agent.destroy(); store.close();
}

(process.argv[2] === '--cfg3' ? cfg3() : classic()).then(() => { Date.now = realNow; }).catch(err => { console.error(err); process.exit(1); });
