'use strict';
// GPU checks of the Node host over the N-API addon: run on an MI355X box with TZ=UTC.
//  1. sessionStore (device-backed) replays the G1-G4 fixture recorded from the real reference module
//     (tests/golden/sessionstore_g1_g4.json) and must answer exactly as the reference did.
//  1b. the same store replays the G5 call trace (tests/golden/sessionstore_g5_trace.json: 1 345 recorded calls into the
//      real module) and must give every answer the reference gave, plus per-user feeds from the device at every census.
//  2. scanFeeds() on the synthetic corpus equals the reference-faithful JS restatement (oracle/ref_faithful.js).
//  3. GET /api/calendar end to end: cookie auth, 401 / 403 / 423, {events} sorted by startTs.
process.env.TZ = 'UTC';
const assert = require('assert');
const fs = require('fs');
const http = require('http');
const path = require('path');

const REPO = path.join(__dirname, '..', '..', '..');
const golden = JSON.parse(fs.readFileSync(path.join(REPO, 'tests', 'golden', 'sessionstore_g1_g4.json'), 'utf8'));
const refJs = require(path.join(REPO, 'oracle', 'ref_faithful.js'));     // the checker, never the product
const {createStore} = require('../sessionStore');
const {createFeedService} = require('../feedService');
const {createServer} = require('../server');
const dc = require('../disciplineConfig');

let checks = 0;
const eq = (a, b, msg) => { assert.deepStrictEqual(a, b, msg); checks++; };
const realNow = Date.now;
let fakeNow = 0;
Date.now = () => fakeNow;

function replayCorpus(){
  const store = createStore();
  const tokens = golden.sessions.map(r => {
    fakeNow = r.createdAt;
    const made = store.createSession(r.user);
    assert.strictEqual(made.expiresAt, r.expiresAt);
    return made.token;
  });
  return {store, tokens};
}

function get(port, cookie){
  return new Promise((resolve, reject) => {
    const req = http.request({host: '127.0.0.1', port, path: '/api/calendar', method: 'GET', headers: cookie ? {cookie} : {}}, res => {
      let body = '';
      res.on('data', c => { body += c; });
      res.on('end', () => resolve({status: res.statusCode, body: JSON.parse(body)}));
    });
    req.on('error', reject);
    req.end();
  });
}

(async () => {
  // ---- 1. G1 liveness
  for(const c of golden.G1){
    const {store, tokens} = replayCorpus();
    fakeNow = c.now;
    eq(tokens.map(t => (store.getSession(t) !== null ? 1 : 0)), c.live, 'G1 now=' + c.now);
    // and the same answer from the device: live rows == rows selected by a scan with no window
    const res = store.scanFeeds({now: c.now});
    const live = new Array(tokens.length).fill(0);
    res.idx.forEach(i => { live[i] = 1; });
    eq(live, c.live, 'G1 device now=' + c.now);
    store.close();
  }
  { const {store} = replayCorpus(); eq([store.getSession(''), store.getSession(null), store.getSession(undefined)], [null, null, null]); store.close(); }
  // ---- G2 purge (device expired-queue scan keeps the host map exact)
  for(const c of golden.G2){
    const {store, tokens} = replayCorpus();
    fakeNow = c.now;
    eq(store.purgeExpiredSessions(), undefined);
    eq(store.size(), c.survivors.length);
    eq(tokens.map((t, i) => (store.getSession(t) !== null ? i : -1)).filter(i => i >= 0), c.survivors, 'G2');
    store.close();
  }
  // ---- G3 deleteSessionsForUser (device user-match scan), incl. unknown and falsy ids
  for(const c of golden.G3){
    const {store, tokens} = replayCorpus();
    fakeNow = c.observe_now;
    eq(store.deleteSessionsForUser(c.user), undefined);
    eq(tokens.map((t, i) => (store.getSession(t) !== null ? i : -1)).filter(i => i >= 0), c.survivors, 'G3 ' + c.user);
    const res = store.scanFeeds({now: c.observe_now});
    eq(Array.from(res.idx).sort((a, b) => a - b), c.survivors, 'G3 device ' + c.user);
    store.close();
  }
  // ---- G4 touch
  {
    const {store, tokens} = replayCorpus();
    for(const t of golden.G4){
      fakeNow = t.now;
      const got = store.touchSession(tokens[t.row]);
      eq(got, t.returned, 'G4 row ' + t.row);
      if(t.after){
        const s = store.getSession(tokens[t.row]);
        eq({userId: s.userId, createdAt: s.createdAt, expiresAt: s.expiresAt}, t.after);
        store.flush();
        eq(Number(store.fetchRows(Int32Array.of(t.row)).end[0]), t.after.expiresAt, 'device end after touch');
      }
    }
    store.close();
  }

  // ---- 1b. G5: a recorded call sequence of the real module, replayed call for call on the device-backed store
  {
    const trace = JSON.parse(fs.readFileSync(path.join(REPO, 'tests', 'golden', 'sessionstore_g5_trace.json'), 'utf8'));
    const store = createStore();
    const tokens = [], userOfRow = [];
    const pub = r => (r === null ? null : {userId: r.userId, createdAt: r.createdAt, expiresAt: r.expiresAt});
    for(const op of trace.ops){
      fakeNow = op.now;
      if(op.op === 'create'){
        const made = store.createSession(op.user);
        eq(made.expiresAt, op.expiresAt);
        tokens.push(made.token); userOfRow.push(op.user);
      } else if(op.op === 'get'){
        eq(pub(store.getSession(tokens[op.tok])), op.result, 'G5 get');
      } else if(op.op === 'touch'){
        const r = store.touchSession(tokens[op.tok]);
        eq(r === null ? null : {userId: r.userId, expiresAt: r.expiresAt}, op.result, 'G5 touch');
      } else if(op.op === 'del'){
        eq(store.deleteSession(tokens[op.tok]), undefined);
      } else if(op.op === 'delUser'){
        eq(store.deleteSessionsForUser(op.user), undefined);
      } else if(op.op === 'purge'){
        eq(store.purgeExpiredSessions(), undefined);
      } else if(op.op === 'census'){
        const live = [];
        tokens.forEach((t, i) => { if(store.getSession(t) !== null){ live.push(i); } });
        eq(live, op.live, 'G5 census now=' + op.now);
        // the batched device scan agrees: selected rows == live rows, grouped per user in row order
        if(tokens.length){
          const res = store.scanFeeds({now: op.now});
          eq(Array.from(res.idx.subarray(0, res.m)).sort((a, b) => a - b), op.live, 'G5 device census');
          const ids = store.userIds();
          for(let u = 0; u < ids.length; u++){
            const rows = Array.from(res.idx.subarray(Number(res.offsets[u]), Number(res.offsets[u + 1])));
            assert.deepStrictEqual(rows, op.live.filter(r => userOfRow[r] === ids[u]));
          }
          checks++;
        }
      }
    }
    eq(tokens.length, trace.sessions);
    store.close();
  }

  // ---- 2. scanFeeds == reference-faithful JS on the synthetic corpus (config 1 shape: 1k sessions / 10 users / 3 disciplines)
  {
    const n = 1000, U = 10, D = 3, T0 = 1700000000000;
    const rows = refJs.genCorpus(0x5EED5EEDn, n, U, D);
    const store = createStore();
    const native = store.native;
    native.genSynthetic(store.ctx, 0x5EED5EEDn, n, 0, n, U, D, 0);
    const s = new BigInt64Array(n), e = new BigInt64Array(n), u = new Int32Array(n), d = new Int32Array(n);
    native.readColumns(store.ctx, s, e, u, d);
    eq(Array.from(s, Number), rows.map(r => r.start)); eq(Array.from(e, Number), rows.map(r => r.end));
    eq(Array.from(u), rows.map(r => r.user)); eq(Array.from(d), rows.map(r => r.disc));
    const sessions = refJs.buildMap(rows);
    for(const [now, cutoff, mask] of [[T0 - 6 * 3600 * 1000, T0 - 61 * 86400 * 1000, 5n], [T0 - 200 * 86400 * 1000, T0 - 61 * 86400 * 1000, 7n], [0, 0, 2n]]){
      native.setDisciplines(store.ctx, mask, D);
      const counts = new Int32Array(U), offsets = new BigInt64Array(U + 1), idx = new Int32Array(n);
      const m = native.scan(store.ctx, now, cutoff, counts, offsets, idx);
      const want = refJs.scanFeeds(sessions, U, now, cutoff, x => x >= 0 && x < D && ((mask >> BigInt(x)) & 1n) === 1n);
      eq(Array.from(counts), want.map(f => f.length));
      eq(Array.from(idx.subarray(0, m)), [].concat.apply([], want.map(f => f.map(x => x.row))));
      eq(Number(offsets[U]), m);
      // async variant gives the same bytes without blocking the event loop
      const c2 = new Int32Array(U), o2 = new BigInt64Array(U + 1), i2 = new Int32Array(n);
      const m2 = await new Promise((res, rej) => native.scanAsync(store.ctx, now, cutoff, c2, o2, i2, (err, mm) => (err ? rej(err) : res(mm))));
      eq([m2, Array.from(c2), Array.from(i2.subarray(0, m2))], [m, Array.from(counts), Array.from(idx.subarray(0, m))]);
    }
    assert.throws(() => native.scan(store.ctx, NaN, 0, new Int32Array(U), new BigInt64Array(U + 1), new Int32Array(n)), /finite integers/);
    checks++;
    store.close();
  }

  // ---- 2b. persistence: save -> close -> restore gives a store that answers like the one that was saved
  {
    const os = require('os');
    const dir = fs.mkdtempSync(path.join(os.tmpdir(), 'pie-store-'));
    const a = createStore();
    const toks = [];
    fakeNow = 1760000000000;
    for(let i = 0; i < 300; i++){ fakeNow += (i % 5 === 0 ? 0 : 1000 * (1 + i % 7)); toks.push(a.createSession('user-' + (i % 17), ['drones', 'audio', 'video'][i % 3]).token); }
    fakeNow += 3600 * 1000;
    a.touchSession(toks[10]); a.touchSession(toks[200]); a.deleteSession(toks[11]); a.deleteSessionsForUser('user-3');
    fakeNow += 9 * 3600 * 1000;                       // some of the early sessions are now close to expiry, none purged yet
    a.save(dir);
    const at = fakeNow + 3 * 3600 * 1000;             // an instant where part of them has expired
    fakeNow = at;
    const wantScan = a.scanFeeds({now: at});
    const wantIdx = Array.from(wantScan.idx), wantCounts = Array.from(wantScan.counts), wantUsers = a.userIds().slice();
    const wantLive = toks.map(t => { const r = a.getSession(t); return r === null ? null : {userId: r.userId, createdAt: r.createdAt, expiresAt: r.expiresAt}; });
    a.close();
    const b = createStore();
    b.restore(dir);
    eq(b.userIds(), wantUsers);
    const gotScan = b.scanFeeds({now: at});
    eq([Array.from(gotScan.idx), Array.from(gotScan.counts)], [wantIdx, wantCounts], 'restored scan');
    eq(toks.map(t => { const r = b.getSession(t); return r === null ? null : {userId: r.userId, createdAt: r.createdAt, expiresAt: r.expiresAt}; }), wantLive, 'restored sessions');
    const fresh = b.createSession('user-1', 'drones');          // the restored table keeps growing
    eq(b.getSession(fresh.token).userId, 'user-1');
    eq(b.scanFeeds({now: at}).m, gotScan.m + 1);
    assert.throws(() => b.restore(dir), /empty store/); checks++;
    b.close();
    fs.readdirSync(dir).forEach(f => fs.unlinkSync(path.join(dir, f))); fs.rmdirSync(dir);
  }

  // ---- 3. HTTP seam
  {
    const store = createStore();
    const users = new Map([
      ['u-lead', {id: 'u-lead', roles: ['drones.lead']}], ['u-crew', {id: 'u-crew', roles: [' Drones.Crew ']}],
      ['u-audio', {id: 'u-audio', roles: ['audio.lead']}], ['u-admin', {id: 'u-admin', roles: ['admin']}],
      ['u-reset', {id: 'u-reset', roles: ['drones.lead'], needsPasswordReset: true}]
    ]);
    const base = 1750000000000;
    const cookies = {};
    let t = base;
    const mk = (uid, disc) => { fakeNow = t; t += 1000; return store.createSession(uid, disc).token; };
    cookies['u-lead'] = mk('u-lead', 'drones');
    mk('u-lead', 'audio'); mk('u-lead', 'drones'); mk('u-crew', 'video');
    fakeNow = t - 1000; mk('u-lead', 'lighting'); t += 1000;                 // equal createdAt as the previous u-lead row? no: distinct rows, tie test below
    cookies['u-crew'] = mk('u-crew', 'drones'); cookies['u-audio'] = mk('u-audio', 'audio');
    cookies['u-admin'] = mk('u-admin', 'drones'); cookies['u-reset'] = mk('u-reset', 'drones');
    const ghost = mk('u-ghost', 'drones');
    fakeNow = t + 5000;
    const feeds = createFeedService(store);
    const server = createServer({store, feeds, findUserById: id => users.get(id) || null});
    await new Promise(r => server.listen(0, '127.0.0.1', r));
    const port = server.address().port;
    const ck = tok => 'theme=dark; ' + store.SESSION_COOKIE_NAME + '=' + tok;
    eq(await get(port, null), {status: 401, body: {error: 'Authentication required'}});
    eq(await get(port, ck('deadbeef')), {status: 401, body: {error: 'Authentication required'}});
    eq(await get(port, ck(ghost)), {status: 401, body: {error: 'Authentication required'}});
    eq(await get(port, ck(cookies['u-audio'])), {status: 403, body: {error: 'Insufficient permissions'}});
    eq(await get(port, ck(cookies['u-reset'])), {status: 423, body: {error: 'Password reset required'}});
    const lead = await get(port, ck(cookies['u-lead']));
    eq(lead.status, 200);
    eq(lead.body.events.length, 4);
    eq(lead.body.events.map(e => e.startTs), lead.body.events.map(e => e.startTs).slice().sort((a, b) => a - b));
    eq(Object.keys(lead.body.events[0]), ['id', 'title', 'description', 'location', 'start', 'end', 'startTs', 'endTs', 'allDay', 'eventName', 'showNumber', 'color']);
    eq(lead.body.events.map(e => e.eventName), ['DRONES', 'AUDIO', 'DRONES', 'LIGHTING']);
    eq(lead.body.events[0].endTs - lead.body.events[0].startTs, store.SESSION_TTL_MS);
    eq((await get(port, ck(cookies['u-crew']))).body.events.length, 2);
    eq((await get(port, ck(cookies['u-admin']))).body.events.length, 1);
    // requests at the same instant on an unchanged store share one device scan; a change to the store or the clock does not
    {
      const before = feeds.scansRun();
      await get(port, ck(cookies['u-lead'])); await get(port, ck(cookies['u-crew'])); await get(port, ck(cookies['u-admin']));
      eq(feeds.scansRun() - before, 0, 'same ms, same store: the scan of the request before them is shared');
      fakeNow += 1;
      const again = await get(port, ck(cookies['u-lead']));
      eq(feeds.scansRun() - before, 1, 'clock moved');
      eq(again.body, lead.body);
      const extra = store.createSession('u-lead', 'drones');
      const more = await get(port, ck(cookies['u-lead']));
      eq(feeds.scansRun() - before, 2, 'store changed');
      eq(more.body.events.length, 5);
      store.deleteSession(extra.token);
      eq((await get(port, ck(cookies['u-lead']))).body, lead.body);
      eq(feeds.scansRun() - before, 3);
    }
    // the same feed as iCalendar text (new route, same auth): one VEVENT per event, same order, UIDs = event ids
    const icsGet = cookie => new Promise((resolve, reject) => {
      http.get({host: '127.0.0.1', port, path: '/api/calendar.ics', headers: cookie ? {cookie} : {}}, res => {
        let b = ''; res.on('data', c => { b += c; }); res.on('end', () => resolve({status: res.statusCode, type: res.headers['content-type'], body: b}));
      }).on('error', reject);
    });
    eq((await icsGet(null)).status, 401);
    eq((await icsGet(ck(cookies['u-audio']))).status, 403);
    const ics = await icsGet(ck(cookies['u-lead']));
    eq([ics.status, ics.type], [200, 'text/calendar; charset=utf-8']);
    eq(ics.body.split('\r\n').filter(l => l.startsWith('UID:')).map(l => l.slice(4)), lead.body.events.map(e => e.id));
    eq(ics.body.split('\r\n').filter(l => l.startsWith('SUMMARY:')).map(l => l.slice(8)), lead.body.events.map(e => e.title));
    const health = await new Promise((resolve, reject) => {
      http.get({host: '127.0.0.1', port, path: '/api/health'}, res => { let b = ''; res.on('data', c => { b += c; }); res.on('end', () => resolve(JSON.parse(b))); }).on('error', reject);
    });
    eq([health.status, health.storage, health.scan.rows, health.scan.users], ['ok', 'MI355X HBM columns', 11, 6]);   // 10 sessions + the one created and deleted above (rows are tombstoned, not removed)
    // expiry: 12 h later every session is dead -> 401; the device agrees (no live rows)
    fakeNow = t + store.SESSION_TTL_MS + 10000;
    eq((await get(port, ck(cookies['u-lead']))).status, 401);
    eq(store.scanFeeds({now: fakeNow}).m, 0);
    // a device failure surfaces as the reference's 500 shape
    const quiet = console.error; console.error = () => {};
    const broken = createServer({store: Object.assign({}, store, {getSession: () => ({userId: 'u-lead'})}),
      feeds: {eventsForUser: () => { throw new Error('device lost'); }}, findUserById: id => users.get(id)});
    await new Promise(r => broken.listen(0, '127.0.0.1', r));
    eq(await get(broken.address().port, ck('x')), {status: 500, body: {error: 'Internal server error', detail: 'device lost'}});
    console.error = quiet;
    server.close(); broken.close(); store.close();
  }
  // ---- 3b. native serialiser: byte-identical to JSON.stringify of the JS-built events, incl. sentinel ends, every
  //          discipline, midnight (allDay) rows; a year beyond 9999 falls back to the JS path
  {
    const cf = require('../calendarFeed');
    const store = createStore();
    const native = store.native;
    const names = dc.DISCIPLINES.map(d => d.name);
    const n = 500;
    const idx = new Int32Array(n), s = new BigInt64Array(n), e = new BigInt64Array(n), d = new Int32Array(n);
    let seed = 12345;
    const rnd = () => { seed = (Math.imul(seed, 1664525) + 1013904223) >>> 0; return seed / 4294967296; };
    for(let i = 0; i < n; i++){
      idx[i] = Math.floor(rnd() * 2e9);
      let st = 1700000000000 - Math.floor(rnd() * 1e10);
      if(i % 7 === 0){ st = Math.floor(st / 86400000) * 86400000; }          // midnight UTC
      s[i] = BigInt(st);
      e[i] = i % 5 === 0 ? cf.END_NONE : BigInt(i % 7 === 0 && i % 2 === 0 ? st + 86400000 : st + 43200000);
      d[i] = i % names.length;
    }
    s[3] = -5000000000000n; e[3] = -4999999999000n;                          // year 1811
    const table = names.map(nm => { const p = cf.eventFromRow(1, 0n, 0n, nm); const j = JSON.stringify(nm); return [j.slice(1, j.length - 1), JSON.stringify(p.eventName), JSON.stringify(p.color)]; });
    const want = JSON.stringify({events: Array.from(idx, (row, i) => cf.eventFromRow(row, s[i], e[i], names[d[i]]))});
    const got = native.serializeEvents(idx, n, s, e, d, table);
    eq(got.toString('utf8') === want, true, 'native serialiser bytes');
    eq(native.serializeEvents(idx, 0, s, e, d, table).toString(), '{"events":[]}');
    // the iCalendar form: native bytes == the JS emitter over the same events
    const summaries = names.map(nm => cf.icsEscape(nm + ' session #'));
    const icsWant = cf.toICalendar(Array.from(idx, (row, i) => cf.eventFromRow(row, s[i], e[i], names[d[i]])), {dtstamp: 1700000000999});
    const icsGot = native.serializeICal(idx, n, s, e, d, summaries, 1700000000999);
    eq(icsGot.toString('utf8') === icsWant, true, 'native iCal bytes');
    eq(native.serializeICal(idx, 0, s, e, d, summaries, 0).toString(), cf.toICalendar([], {dtstamp: 0}));
    eq(native.serializeICal(idx, 3, s, e, d, names.map(() => 'x'.repeat(70)), 0), null);   // would need folding -> JS path
    s[0] = 300000000000000n;                                                 // year 11476: not covered -> null
    eq(native.serializeEvents(idx, n, s, e, d, table), null);
    eq(native.serializeICal(idx, n, s, e, d, summaries, 0), null);
    d[1] = 99;
    eq(native.serializeEvents(idx, 2, s.subarray(1), e.subarray(1), d.subarray(1), table), null);
    store.close();
  }

  // ---- 4. expired-session dispatch queue: device-ordered, drained sequentially, failures summarised
  {
    const {dispatchExpiredSessions} = require('../dispatchQueue');
    const store = createStore();
    const base = 1760000000000;
    const made = [];
    for(let i = 0; i < 40; i++){ fakeNow = base + i * 60000; store.createSession('user-' + (i % 7), i % 2 ? 'audio' : 'drones'); made.push(fakeNow + store.SESSION_TTL_MS); }
    const prev = made[9], now = made[24];                      // rows 10..24 expire in (prev, now]
    const seen = [];
    let inFlight = 0, maxInFlight = 0;
    const send = async (payload, meta) => {
      inFlight++; maxInFlight = Math.max(maxInFlight, inFlight);
      await new Promise(r => setImmediate(r));
      seen.push(payload.sessionRow); inFlight--;
      if(payload.sessionRow === 13){ return {success: false, error: 'HTTP 500'}; }
      if(payload.sessionRow === 17){ throw new Error('socket hang up'); }
      assert.strictEqual(meta.event, 'session.expired');
      return {success: true};
    };
    const summary = await dispatchExpiredSessions(store, prev, now, send);
    eq(seen, Array.from({length: 15}, (_, k) => 10 + k));      // ascending row order
    eq(maxInFlight, 1);                                        // strictly sequential, like the reference's await loop
    eq([summary.success, summary.dispatched, summary.failed, summary.total], [false, 13, 2, 15]);
    eq(summary.error, 'One or more expired-session payloads failed to dispatch');
    eq(summary.results[7], {success: false, error: 'socket hang up', sessionRow: 17});
    const p0 = summary.results.length && (await dispatchExpiredSessions(store, prev, prev, send));
    eq(p0, {success: true, dispatched: 0, failed: 0, total: 0, results: []});
    const one = [];
    await dispatchExpiredSessions(store, made[38], made[39], async p => { one.push(p); return {success: true}; });
    eq(one, [{sessionRow: 39, userId: 'user-4', discipline: 'audio', createdAt: new Date(base + 39 * 60000).toISOString(),
      expiredAt: new Date(made[39]).toISOString()}]);
    // archive chain: users u0..u6 first appear in rows 0..6; at `t` only users whose FIRST session is >= 12 h old qualify
    const {dispatchArchivedGroups} = require('../dispatchQueue');
    const order = [];
    const t = base + 3 * 60000 + store.SESSION_TTL_MS;          // earliest of user-0..user-3 (rows 0..3) is old enough
    const sum2 = await dispatchArchivedGroups(store, t, async p => { order.push(p.sessionRow); return {success: true}; });
    const expectRows = [];
    for(let g = 0; g < 4; g++){ for(let i = g; i < 40; i += 7){ expectRows.push(i); } }
    eq(order, expectRows);
    eq([sum2.success, sum2.total], [true, expectRows.length]);
    store.close();
  }
  // ---- 5. rows are reclaimed; a session that getSession() found expired still reaches the dispatch queue
  {
    const store = createStore({compactMinRows: 64});
    const base = 1770000000000;
    const toks = [];
    for(let i = 0; i < 200; i++){ fakeNow = base + i * 1000; toks.push(store.createSession('u' + (i % 9)).token); }
    store.flush();
    eq(store.tableRows(), 200);
    // sessions 0..9 expire and are looked up (the lookup drops them from the map, as the reference's getSession does)
    fakeNow = base + 9 * 1000 + store.SESSION_TTL_MS;
    for(let i = 0; i < 10; i++){ eq(store.getSession(toks[i]), null); }
    eq(store.size(), 190);
    eq(Array.from(store.expiredRows(null, fakeNow)), [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]);   // still in the queue: `end` kept
    // explicit deletes and a user delete retire rows; once they outweigh the live rows the table is compacted
    for(let i = 10; i < 120; i++){ store.deleteSession(toks[i]); }
    eq(store.compactions(), 0);
    store.purgeExpiredSessions();                                  // its flush compacts (110 of 200 rows are gone), then it
    eq(store.compactions(), 1);                                    // reports + retires the ten expired rows
    eq(store.tableRows(), 90);
    const before = store.scanFeeds({now: fakeNow});
    eq(before.m, 80);
    store.createSession('fresh');
    const after = store.scanFeeds({now: fakeNow});
    eq(store.tableRows(), 91);
    eq(after.m, 81);
    eq(Array.from(store.expiredRows(null, fakeNow)), [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]);   // the queue itself forgets nothing
    for(let i = 120; i < 200; i++){ const sess = store.getSession(toks[i]); eq(sess !== null && sess.userId === 'u' + (i % 9), true); }
    for(let i = 0; i < 120; i += 17){ eq(store.getSession(toks[i]), null); }
    // feeds after compaction: every live session of u3, ascending by createdAt
    const feeds = createFeedService(store).allFeeds({now: fakeNow, cutoff: 0});
    eq(feeds.get('u3').map(ev => ev.startTs), toks.map((t, i) => i).filter(i => i >= 120 && i % 9 === 3).map(i => base + i * 1000));
    store.close();
  }

  // ---- 6. the requests of one turn, ONE table pass: batched bodies == per-request bodies; HTTP coalescing
  {
    const store = createStore();
    const base = 1780000000000;
    const toks = {};
    for(let i = 0; i < 3000; i++){
      fakeNow = base + i * 977;
      const uid = 'user-' + (i % 37);
      const t = store.createSession(uid, dc.DISCIPLINES[i % dc.DISCIPLINES.length].id).token;
      if(toks[uid] === undefined){ toks[uid] = t; }
    }
    const feedsA = createFeedService(store), feedsB = createFeedService(store);
    const t1 = base + 3000 * 977 + 5;
    const requests = [];
    for(let k = 0; k < 150; k++){
      requests.push({userId: 'user-' + (k % 37), query: {now: t1 + (k % 19) * 1000, cutoff: base + (k % 3) * 100000, disciplines: k % 4 === 0 ? ['drones', 'audio'] : undefined}});
    }
    requests.push({userId: 'nobody', query: {now: t1, cutoff: 0}});
    fakeNow = t1;
    const bodies = feedsA.eventsJsonForRequests(requests);
    eq(feedsA.batchesRun() >= 2, true);                            // 150 distinct (now, cutoff, disciplines) keys, at most 64 per batch
    requests.forEach((r, i) => { eq(bodies[i].equals(feedsB.eventsJsonForUser(r.userId, r.query)), true, 'request ' + i); });
    eq(bodies[150].toString(), '{"events":[]}');
    // over HTTP: concurrent requests are answered from one batch
    const users = new Map();
    for(let u = 0; u < 37; u++){ users.set('user-' + u, {id: 'user-' + u, roles: ['drones.crew']}); }
    const feedsC = createFeedService(store);
    const server = createServer({store, findUserById: id => users.get(id) || null, feeds: feedsC, coalesce: true,
      query: () => ({now: t1, cutoff: base})});
    await new Promise(r => server.listen(0, '127.0.0.1', r));
    const port = server.address().port;
    const names = Array.from(users.keys()).slice(0, 12);
    const answers = await Promise.all(names.map(nm => get(port, 'mt_session=' + toks[nm])));
    const plain = createFeedService(store);
    answers.forEach((a, i) => {
      eq(a.status, 200);
      eq(JSON.stringify(a.body), plain.eventsJsonForUser(names[i], {now: t1, cutoff: base}).toString());
    });
    eq(feedsC.batchesRun() >= 1 && feedsC.batchesRun() <= 12, true);
    eq((await get(port, null)).status, 401);
    await new Promise(r => server.close(r));
    // CSV of a queue: native bytes == the JS builders
    const dq = require('../dispatchQueue');
    const rowsQ = store.expiredRows(null, base + 500 * 977 + store.SESSION_TTL_MS);
    eq(rowsQ.length, 501);
    const cols = store.fetchRows(rowsQ);
    const lines = [dq.EXPORT_COLUMNS.join(',')];
    for(let i = 0; i < rowsQ.length; i++){ lines.push(dq.buildCsvRow(dq.buildTableRow(dq.buildExpiredSessionPayload(rowsQ[i], cols, i, store.userIds())))); }
    eq(dq.queueCsv(store, rowsQ).toString('utf8') === lines.join('\n'), true, 'native CSV bytes');
    const odd = ['plain', 'comma, inside', 'quote " inside', 'new\nline', ''];
    const got = store.native.serializeCsv(Int32Array.from([5, 6, 7, 8, 9]), 5, cols.start, cols.end, Int32Array.from([0, 1, 2, 3, 4]), cols.disc, odd,
      dc.DISCIPLINES.map(d => d.id), dq.EXPORT_COLUMNS.join(','));
    const wantOdd = [dq.EXPORT_COLUMNS.join(',')];
    for(let i = 0; i < 5; i++){ wantOdd.push(dq.buildCsvRow(dq.buildTableRow(dq.buildExpiredSessionPayload(5 + i, {start: cols.start, end: cols.end, user: Int32Array.from([0, 1, 2, 3, 4]), disc: cols.disc}, i, odd)))); }
    eq(got.toString('utf8'), wantOdd.join('\n'));
    store.close();
  }

  // ---- 7. the communicator through the addon (one GPU on this box: a world of one shard): batch + exchange == single scans
  {
    const native = require('../pieNative').load();
    const comm = native.commCreate(Int32Array.from([0]));
    eq(native.commWorld(comm), 1);
    const n = 300000, U = 2000, D = 7;
    native.commGenSyntheticSharded(comm, 0x5EED5EED, n, U, D, 0);
    const shard = native.commCtx(comm, 0);
    native.setDisciplines(shard, 127n, D);
    const T0 = 1700000000000;
    const nows = BigInt64Array.from([BigInt(T0 - 6 * 3600000), BigInt(T0 - 7 * 3600000), BigInt(T0 - 30 * 86400000)]);
    const cuts = BigInt64Array.from([BigInt(T0 - 61 * 86400000), 0n, BigInt(T0 - 40 * 86400000)]);
    const masks = BigUint64Array.from([0x55n, 0x7Fn, 0x2An]);
    const ms = native.commScanBatchGather(comm, nows, cuts, masks);
    const uPad = native.commUPad(comm, 0);
    eq(uPad, U);
    const single = native.ctxCreate(0);
    native.genSynthetic(single, 0x5EED5EED, n, 0, n, U, D, 0);
    for(let q = 0; q < 3; q++){
      native.setDisciplines(single, masks[q], D);
      const counts = new Int32Array(U), offsets = new BigInt64Array(U + 1), idx = new Int32Array(n);
      const m = native.scan(single, nows[q], cuts[q], counts, offsets, idx);
      eq(ms[0][q], m);
      const off = new Int32Array(uPad + 1), rows = new Int32Array(Math.max(m, 1));
      eq(native.commReadGathered(comm, 0, 0, q, off, rows), m);
      eq(Array.from(off), Array.from(offsets, Number));
      eq(Buffer.from(rows.buffer, 0, m * 4).equals(Buffer.from(idx.buffer, 0, m * 4)), true);
    }
    // handles: a destroyed context throws instead of touching freed memory; short output arrays are refused
    assert.throws(() => native.scan(single, nows[0], cuts[0], new Int32Array(U - 1), new BigInt64Array(U + 1), new Int32Array(n)), /Int32Array\[>= users\]/);
    assert.throws(() => native.scan(single, nows[0], cuts[0], new Int32Array(U), new BigInt64Array(U), new Int32Array(n)), /users \+ 1/);
    native.ctxDestroy(single);
    assert.throws(() => native.scanDevice(single, nows[0], cuts[0]), e => e.code === -6);
    checks += 3;
    // the pipelined union exchange through the addon: reserve, two steps begun ahead, finish / collect, every query's feed a
    // filter of the gathered union; a reservation that is too small is reported at collect (code -5) with what to reserve
    {
      const single2 = native.ctxCreate(0);
      native.genSynthetic(single2, 0x5EED5EED, n, 0, n, U, D, 0);
      native.commStepReserve(comm, 3, 8);
      native.commStepBegin(comm, nows, cuts, masks);
      native.commStepFinish(comm);
      assert.throws(() => native.commStepCollect(comm), e => e.code === -5);
      const need = native.commNeededCap(comm);
      eq(need > 8, true);
      native.commStepReserve(comm, 3, need);
      const steps = [];
      native.commStepBegin(comm, nows, cuts, masks);
      for(let i = 0; i < 5; i++){
        if(i + 1 < 5){ native.commStepBegin(comm, nows, cuts, masks); }
        const msq = native.commStepFinish(comm);
        eq(msq[0].length, 3);
        if(i >= 1){ steps.push(native.commStepCollect(comm)); }
      }
      steps.push(native.commStepCollect(comm));
      eq(steps, [1, 2, 3, 4, 5]);
      const up = native.commStepUPad(comm, 0, 5);
      const uoff = new Int32Array(up + 1), rows = new Int32Array(need), um = new BigUint64Array(need);
      const mu = native.commStepReadGathered(comm, 0, 0, 5, uoff, rows, um);
      eq(uoff[U], mu);
      for(let q = 0; q < 3; q++){
        native.setDisciplines(single2, masks[q], D);
        const counts = new Int32Array(U), offsets = new BigInt64Array(U + 1), idx = new Int32Array(n);
        const m = native.scan(single2, nows[q], cuts[q], counts, offsets, idx);
        const mine = [];
        for(let i = 0; i < mu; i++){ if((um[i] >> BigInt(q)) & 1n){ mine.push(rows[i]); } }
        eq(mine.length, m, 'union query ' + q);
        eq(Buffer.from(Int32Array.from(mine).buffer).equals(Buffer.from(idx.buffer, 0, m * 4)), true, 'union rows of query ' + q);
      }
      native.ctxDestroy(single2);
      checks += 4;
    }
    native.commDestroy(comm);
    assert.throws(() => native.commWorld(comm), /communicator/);
    // ADVICE r02: a context handle handed out by commCtx must not outlive its communicator as a dangling pointer
    assert.throws(() => native.scanDevice(shard, nows[0], cuts[0]), e => e.code === -6);
    assert.throws(() => native.stats(shard), e => e.code === -6);
    checks += 3;
  }
  // ---- 8. the ordered run behind the store: same feeds as the general path, on a skewed table, after touches and deletes
  {
    const base = 1790000000000;
    const mk = orderedRun => {
      const st = createStore({orderedRun});
      const toks = [];
      for(let i = 0; i < 4000; i++){
        fakeNow = base + i * 1013;
        const uid = i % 3 === 0 ? 'head' : 'user-' + (i % 41);          // one user owns a third of the sessions
        toks.push(st.createSession(uid, dc.DISCIPLINES[i % dc.DISCIPLINES.length].id).token);
      }
      return {st, toks};
    };
    const a = mk(2), b = mk(0);
    const t1 = base + 4000 * 1013 + 7;
    const same = (tag) => {
      const ra = a.st.scanFeeds({now: t1, cutoff: base + 500000}), rb = b.st.scanFeeds({now: t1, cutoff: base + 500000});
      eq(ra.m, rb.m, tag + ' m');
      eq(Array.from(ra.counts), Array.from(rb.counts), tag + ' counts');
      eq(Array.from(ra.idx.subarray(0, ra.m)), Array.from(rb.idx.subarray(0, rb.m)), tag + ' idx');
      eq((a.st.native.stats(a.st.ctx).k1Variant & 0x2000) !== 0, true, tag + ' ran on the ordered run');
      eq((b.st.native.stats(b.st.ctx).k1Variant & 0x2000) === 0, true, tag + ' general path');
    };
    fakeNow = t1;
    same('fresh');
    for(const s of [a, b]){
      for(let i = 0; i < 4000; i += 7){ s.st.touchSession(s.toks[i]); }
      for(let i = 3; i < 4000; i += 11){ s.st.deleteSession(s.toks[i]); }
      s.st.deleteSessionsForUser('user-5');
    }
    same('touched and deleted');
    a.st.close();
    b.st.close();
  }
  // ---- 9. batch lanes behind the store (createStore({batchLanes})): batches of consecutive turns run on different streams of
  //         the context; every request's feed equals the one-lane store's
  {
    const base = 1790100000000;
    const mk = batchLanes => {
      const st = createStore({batchLanes});
      for(let i = 0; i < 3000; i++){
        fakeNow = base + i * 997;
        st.createSession('user-' + (i % 53), dc.DISCIPLINES[i % dc.DISCIPLINES.length].id);
      }
      return st;
    };
    const a = mk(4), b = mk(1);
    eq(a.native.setBatchLanes(a.ctx, 4), 4, 'four lanes');
    const t1 = base + 3000 * 997 + 5;
    fakeNow = t1;
    for(let turn = 0; turn < 6; turn++){                 // six batches: lanes 0, 1, 2, 3, 0, 1
      const qs = [];
      for(let q = 0; q < 5 + turn; q++){ qs.push({now: t1 - 40000000 + 1000 * q + turn, cutoff: base + 200000 * (q % 3), disciplines: q % 2 ? undefined : [dc.DISCIPLINES[q % dc.DISCIPLINES.length].id]}); }
      const ma = a.scanBatchDevice(qs), mb = b.scanBatchDevice(qs);
      eq(Array.from(ma, Number), Array.from(mb, Number), 'turn ' + turn + ' selected rows per query');
      for(const u of [0, 7, 52]){
        for(const qi of [0, qs.length - 1]){
          eq(Array.from(a.batchUserFeed(qi, u)), Array.from(b.batchUserFeed(qi, u)), 'turn ' + turn + ' feed of user ' + u + ' query ' + qi);
        }
      }
    }
    a.close();
    b.close();
  }
  Date.now = realNow;
  console.log('host gpu_test ok: ' + checks + ' checks');
})().catch(err => { console.error(err); process.exit(1); });
