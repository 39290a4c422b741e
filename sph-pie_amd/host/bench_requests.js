'use strict';
// End-to-end request rate of the feed path on one Node thread: N users x S sessions loaded into the device-backed store,
// then R requests (GET /api/calendar over a real loopback socket, round-robin users, the clock ticking 1 ms per request
// so that NO two requests share a scan).  Prints one JSON line.  usage: node bench_requests.js [users] [sessionsPerUser] [requests]
process.env.TZ = 'UTC';
const http = require('http');
const {createStore} = require('./sessionStore');
const {createFeedService} = require('./feedService');
const {createServer} = require('./server');

const U = Number(process.argv[2] || 2000), S = Number(process.argv[3] || 50), R = Number(process.argv[4] || 2000);
const realNow = Date.now;
let fakeNow = 1750000000000;
Date.now = () => fakeNow;

(async () => {
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
  Date.now = realNow;
})().catch(err => { console.error(err); process.exit(1); });
