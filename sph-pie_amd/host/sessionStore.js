'use strict';
// Device-backed session store with the reference module's interface
// (/root/reference/server/sessionStore.js:75-84: createSession, getSession, touchSession, deleteSession,
// deleteSessionsForUser, purgeExpiredSessions, SESSION_TTL_MS, SESSION_COOKIE_NAME — synchronous, never
// throwing on a miss, null for "no session").
//
// What moved to the GPU: the session records live as SoA columns in HBM (int64 start/end, int32 user/disc);
// the two full-table loops of the reference — deleteSessionsForUser (:55-64) and purgeExpiredSessions
// (:66-73) — are device scans that hand back the affected row indices; the per-user feed aggregation is
// scanFeeds().  What stays on the host: the token -> row map (O(1) lookups, sha256) and a mirror of each row's
// {userId, createdAt, expiresAt}, kept exact by the row lists the device scans return, so getSession never
// touches the device.  Mutations are batched and flushed to the device before any scan.
//
// Rows are reclaimed: the reference does sessions.delete(); here a deleted / purged session first becomes a tombstone
// (its row index may still sit in a result the caller holds), and once the dead rows outweigh the live ones the table is
// compacted — live rows re-uploaded in order, the token map re-pointed — so N follows the live sessions, not every login
// ever made.  A session that getSession() finds expired keeps its real `end` on the device (it is dead for every scan
// anyway) so that the expired-session dispatch queue still reports it.
const crypto = require('crypto');
const fs = require('fs');
const path = require('path');
const disciplineConfig = require('./disciplineConfig');
const pieNative = require('./pieNative');

const SESSION_TTL_MS = 12 * 60 * 60 * 1000;
const SESSION_COOKIE_NAME = 'mt_session';
const END_NONE = -(2n ** 63n);          // PIE_END_NONE: tombstone, never live

const sha256hex = text => crypto.createHash('sha256').update(text).digest('hex');

function createStore(options){
  const opts = options || {};
  const native = pieNative.load();                 // throws if the addon / HIP library is missing
  const ctx = native.ctxCreate(opts.device === undefined ? 0 : opts.device);   // throws without a GPU
  // the ordered run (pie_set_ordered_run): 0 never, 1 where the general path is weak (the library's default), 2 always
  if(opts.orderedRun !== undefined){ native.setOrderedRun(ctx, opts.orderedRun); }
  // lanes of the batched scan (pie_set_batch_lanes): 1..4 independent streams, 0 / absent = chosen by table size
  if(opts.batchLanes !== undefined){ native.setBatchLanes(ctx, opts.batchLanes); }

  const rowOfToken = new Map();                    // tokenHash -> row
  const rows = [];                                 // row -> {tokenHash|null, userId, createdAt, expiresAt}
  const userIndex = new Map();                     // userId string -> dense int32
  const userIds = [];                              // dense int32 -> userId string
  let uploaded = 0;                                // rows [0, uploaded) are resident on the device
  let pendingEnd = new Map();                      // row -> BigInt new end (touch / delete of resident rows)
  let lastPurge = null;
  let out = null;                                  // scan output arrays, sized lazily
  let generation = 0;                              // bumped by every change a scan could see AND by every native call that
                                                   // drops the device's last scan result (see feedService's scan sharing)
  let nGone = 0;                                   // rows that no scan, queue or lookup will ever need again
  let listBuf = new Int32Array(1024);              // reused row-list buffer of the device scans (grown, never per call)
  const compactMinRows = opts.compactMinRows === undefined ? 4096 : opts.compactMinRows;
  // A resident BASE of anonymous rows (opts.base = {rows, users, disc, seed?, flags?}): the synthetic corpus of SURVEY.md 8d
  // generated on the device (pie_gen_synthetic), users 'user-0' .. pre-registered — the way a 10^8-session table exists at all
  // on one Node process (the host keeps a map entry per session it ISSUED, not per row of history).  Sessions created through
  // this module are appended behind the base; device row = base + host row.  The full-table list operations that report rows
  // back to the host map (deleteSessionsForUser, purges, compaction, save) are not offered on a based store.
  let base = 0;
  if(opts.base){
    const b = opts.base;
    for(let u = 0; u < b.users; u++){ denseUserEarly('user-' + u); }
    native.genSynthetic(ctx, b.seed === undefined ? 0x5EED5EED : b.seed, b.rows, 0, b.rows, b.users, b.disc, b.flags || 0);
    base = b.rows;
  }
  function denseUserEarly(userId){ userIndex.set(userId, userIds.length); userIds.push(userId); }
  const noBase = what => { if(base > 0){ throw new Error(what + ' is not offered on a store with a synthetic base'); } };

  function rowList(){
    if(listBuf.length < rows.length){
      listBuf = new Int32Array(Math.max(rows.length, 2 * listBuf.length));
    }
    return listBuf;
  }

  function denseUser(userId){
    let u = userIndex.get(userId);
    if(u === undefined){
      u = userIds.length;
      userIndex.set(userId, u);
      userIds.push(userId);
    }
    return u;
  }

  // push host-side mutations to the device: appended rows first, then end updates, so a row that was created
  // and touched in the same batch ends with the touched value
  function flush(){
    if(base === 0 && rows.length >= compactMinRows && nGone * 2 > rows.length){
      compact();
    }
    const k = rows.length - uploaded;
    if(k > 0 || (uploaded === 0 && base === 0)){
      const s = new BigInt64Array(k), e = new BigInt64Array(k);
      const u = new Int32Array(k), d = new Int32Array(k);
      for(let i = 0; i < k; i++){
        const r = rows[uploaded + i];
        s[i] = BigInt(r.createdAt);
        e[i] = r.gone ? END_NONE : BigInt(r.expiresAt);   // a deleted row is a tombstone; an expired one keeps its `end`
        u[i] = r.user;
        d[i] = r.disc;
        pendingEnd.delete(uploaded + i);
      }
      const nUsers = Math.max(userIds.length, 1);
      if(uploaded === 0 && base === 0){
        native.loadColumns(ctx, s, e, u, d, nUsers);
      }else{
        native.appendRows(ctx, s, e, u, d, nUsers);
      }
      uploaded = rows.length;
    }
    if(pendingEnd.size > 0){
      const r = new Int32Array(pendingEnd.size), v = new BigInt64Array(pendingEnd.size);
      let i = 0;
      pendingEnd.forEach((val, row) => { r[i] = base + row; v[i] = val; i++; });
      native.setEnd(ctx, r, v);
      pendingEnd = new Map();
    }
  }

  function forget(row){
    const r = rows[row];
    generation++;
    if(r.tokenHash !== null){
      rowOfToken.delete(r.tokenHash);
      r.tokenHash = null;
    }
  }

  // the row has been reported by a purge / user delete, or was deleted explicitly: nothing will ask for it again
  function retire(row){
    const r = rows[row];
    if(!r.gone){
      r.gone = true;
      nGone++;
    }
  }

  function tombstone(row){
    forget(row);
    retire(row);
    if(row < uploaded){
      pendingEnd.set(row, END_NONE);
    }
  }

  // Rebuild the table from the rows that are still needed (live sessions, and expired ones no purge has reported yet),
  // in their old order; row indices change, so the device's last result is dropped with the generation bump.
  function compact(){
    const keep = [];
    for(let i = 0; i < rows.length; i++){
      if(!rows[i].gone){ keep.push(rows[i]); }
    }
    rows.length = 0;
    rowOfToken.clear();
    for(const r of keep){
      if(r.tokenHash !== null){ rowOfToken.set(r.tokenHash, rows.length); }
      rows.push(r);
    }
    uploaded = 0;                       // the next lines of flush() upload the whole (smaller) table
    pendingEnd = new Map();
    nGone = 0;
    out = null;
    generation++;
    compactions++;
  }
  let compactions = 0;

  // createSession(userId[, disciplineId]) -> {token, expiresAt}.  The second argument is [DERIVED] (the reference
  // record has no discipline field): an id of disciplineConfig.DISCIPLINES, default = the default discipline.
  function createSession(userId, disciplineId){
    const token = crypto.randomBytes(48).toString('hex');
    const tokenHash = sha256hex(token);
    const now = Date.now();
    const expiresAt = now + SESSION_TTL_MS;
    let disc = disciplineId === undefined
      ? (disciplineConfig.DEFAULT_DISCIPLINE ? disciplineConfig.disciplineIndex(disciplineConfig.DEFAULT_DISCIPLINE.id) : 0)
      : disciplineConfig.disciplineIndex(disciplineId);
    rows.push({tokenHash, userId, user: denseUser(userId), disc, createdAt: now, expiresAt});
    generation++;
    rowOfToken.set(tokenHash, rows.length - 1);
    return {token, expiresAt};
  }

  function getSession(token){
    if(!token){
      return null;
    }
    const tokenHash = sha256hex(token);
    const row = rowOfToken.get(tokenHash);
    if(row === undefined){
      return null;
    }
    const r = rows[row];
    if(r.expiresAt <= Date.now()){                  // dead iff expiresAt <= now (:30)
      forget(row);                                  // sessions.delete(hash): the lookup is gone; the row keeps its real
      return null;                                  // `end` on the device, so the expired-session queue still reports it
    }
    return {userId: r.userId, createdAt: r.createdAt, expiresAt: r.expiresAt, tokenHash};
  }

  function touchSession(token){
    const existing = getSession(token);
    if(!existing){
      return null;
    }
    const row = rowOfToken.get(existing.tokenHash);
    const newExpires = Date.now() + SESSION_TTL_MS;
    rows[row].expiresAt = newExpires;
    generation++;
    if(row < uploaded){
      pendingEnd.set(row, BigInt(newExpires));
    }
    return {userId: existing.userId, expiresAt: newExpires};
  }

  function deleteSession(token){
    if(!token){
      return;
    }
    const row = rowOfToken.get(sha256hex(token));
    if(row !== undefined){
      tombstone(row);
    }
  }

  // full-table scan on the device (user == x, strict); the returned row list keeps the host map exact
  function deleteSessionsForUser(userId){
    if(!userId){
      return;
    }
    noBase('deleteSessionsForUser');
    const u = userIndex.get(userId);
    if(u === undefined){
      return;
    }
    flush();
    const list = rowList();
    const k = native.deleteUser(ctx, u, list);
    generation++;                                   // the call drops the device's last scan result even when k == 0
    for(let i = 0; i < k; i++){
      forget(list[i]);
      retire(list[i]);
    }
  }

  // full-table scan on the device; `now` is sampled once (:67).  The device returns the rows that died since the
  // previous purge (prev < end <= now); rows that died earlier were already dropped from the map.
  function purgeExpiredSessions(){
    noBase('purgeExpiredSessions');
    const now = Date.now();
    flush();
    const list = rowList();
    const prev = lastPurge === null ? END_NONE : BigInt(lastPurge);
    const k = native.expiredQueue(ctx, prev, now, list);
    generation++;
    for(let i = 0; i < k; i++){
      forget(list[i]);
      retire(list[i]);
    }
    lastPurge = now;
    return undefined;
  }

  // Retention purge with calendar-month arithmetic (SURVEY 8f-2; /root/reference/server/storage/sqlProvider.js:863-890, :991-1009:
  // a record goes once now >= addMonths(createdAt, months), addMonths = `setMonth(getMonth() + months)` on a LOCAL Date).
  // The device does the month arithmetic for every row under this process's time zone, handed over as a transition table
  // (tzTable.js: built from the engine's own zone rules, so daylight saving moves the result exactly as it does for the
  // reference's Date).  -> number of sessions dropped
  function purgeRetention(months, now){
    noBase('purgeRetention');
    const tz = require('./tzTable').defaultTzTable();
    flush();
    const list = rowList();
    const k = native.retentionPurgeTz(ctx, now === undefined ? Date.now() : now, months === undefined ? 2 : months, tz.transitions, tz.offsets, list);
    generation++;
    for(let i = 0; i < k; i++){
      forget(list[i]);
      retire(list[i]);
    }
    return k;
  }

  // ---- the batched feed scan (replaces the per-request loop) ------------------------------------------
  // -> {counts Int32Array[U], offsets BigInt64Array[U+1], idx Int32Array[M], m, userIds}
  function scanFeeds(query){
    noBase('scanFeeds (the whole result on the host)');
    const q = query || {};
    const now = q.now === undefined ? Date.now() : q.now;
    const cutoff = q.cutoff === undefined ? END_NONE : q.cutoff;
    flush();
    const U = Math.max(userIds.length, 1);
    if(out === null || out.counts.length !== U || out.idx.length < rows.length){
      out = {counts: new Int32Array(U), offsets: new BigInt64Array(U + 1), idx: new Int32Array(Math.max(rows.length, 1))};
    }
    native.setDisciplines(ctx, disciplineConfig.disciplineMask(q.disciplines), disciplineConfig.DISCIPLINES.length);
    const m = native.scan(ctx, now, cutoff, out.counts, out.offsets, out.idx);
    return {counts: out.counts, offsets: out.offsets, idx: out.idx.subarray(0, m), m, userIds};
  }

  // the same scan with its result left in HBM; userFeed(u) then reads one user's rows (two small copies).  This is the
  // per-request path: a request needs one slice, not the counts / offsets / idx of every user.
  function scanDevice(query){
    const q = query || {};
    const now = q.now === undefined ? Date.now() : q.now;
    const cutoff = q.cutoff === undefined ? END_NONE : q.cutoff;
    flush();
    native.setDisciplines(ctx, disciplineConfig.disciplineMask(q.disciplines), disciplineConfig.DISCIPLINES.length);
    return native.scanDevice(ctx, now, cutoff);
  }
  var feedBuf = null;
  function userFeed(u){
    const want = Math.max(base > 0 ? 65536 : rows.length, 1);
    if(feedBuf === null || feedBuf.length < want){
      feedBuf = new Int32Array(want);
    }
    const k = native.userFeed(ctx, u, feedBuf);
    return feedBuf.slice(0, k);
  }

  // ordered device queue of rows with prevNow < expiresAt <= now (no change to the host map: purgeExpiredSessions does that)
  function expiredRows(prevNow, now){
    noBase('expiredRows');
    flush();
    const list = rowList();
    const prev = prevNow === null || prevNow === undefined ? END_NONE : prevNow;
    const k = native.expiredQueue(ctx, prev, now, list);
    generation++;
    return list.slice(0, k);
  }

  // the reference's archive chain on the session table (sqlProvider.js:758-816): rows of every user whose earliest
  // session is at least windowMs old, users in order of first appearance, rows in table order
  function archivedRows(now, windowMs){
    noBase('archivedRows');
    flush();
    const list = rowList();
    const k = native.archiveQueue(ctx, now, windowMs === undefined ? SESSION_TTL_MS : windowMs, list);
    generation++;
    return list.slice(0, k);
  }

  // ---- batched scan: the feed requests of one event-loop turn, ONE table pass (pie_scan_batch) ------------------------
  // queries: up to native batch size of {now, cutoff, disciplines}; -> selected rows per query.  batchUserFeed(qi, u) then
  // reads one user's rows of query qi (two small copies), like userFeed for a single scan.
  const BATCH_MAX = 64;
  function scanBatchDevice(queries){
    if(queries.length < 1 || queries.length > BATCH_MAX){ throw new Error('a batch holds 1..' + BATCH_MAX + ' queries'); }
    flush();
    const nows = new BigInt64Array(queries.length), cutoffs = new BigInt64Array(queries.length);
    const masks = new BigUint64Array(queries.length);
    queries.forEach((q, k) => {
      nows[k] = BigInt(q.now === undefined ? Date.now() : q.now);
      cutoffs[k] = q.cutoff === undefined ? END_NONE : BigInt(q.cutoff);
      masks[k] = BigInt.asUintN(64, BigInt(disciplineConfig.disciplineMask(q.disciplines)));
    });
    native.setDisciplines(ctx, (1n << BigInt(disciplineConfig.DISCIPLINES.length)) - 1n, disciplineConfig.DISCIPLINES.length);
    generation++;                                   // a single-scan result on the device is gone
    return native.scanBatch(ctx, nows, cutoffs, masks);
  }
  function batchUserFeed(qi, u){
    const want = Math.max(base > 0 ? 4096 : rows.length, 1);
    if(feedBuf === null || feedBuf.length < want){
      feedBuf = new Int32Array(want);
    }
    const k = native.batchUserFeed(ctx, qi, u, feedBuf);
    return feedBuf.slice(0, k);
  }
  // the feeds of MANY (query, user) requests of the last batch in one native call (pie_batch_fetch_requests): -> {off, idx,
  // start, end, disc}; request i's rows are idx[off[i] .. off[i + 1]) in feed order with their columns, ready to serialise
  let fetchBufs = null;
  function batchFetch(qis, users){
    const n = qis.length;
    let cap = fetchBufs === null ? Math.max(1024, 8 * n) : fetchBufs.idx.length;
    for(;;){
      if(fetchBufs === null || fetchBufs.idx.length < cap || fetchBufs.off.length < n + 1){
        fetchBufs = {off: new BigInt64Array(Math.max(n + 1, fetchBufs === null ? 0 : fetchBufs.off.length)), idx: new Int32Array(cap),
          start: new BigInt64Array(cap), end: new BigInt64Array(cap), disc: new Int32Array(cap)};
      }
      try{
        const total = native.batchFetchRequests(ctx, qis, users, fetchBufs.off, fetchBufs.idx, fetchBufs.start, fetchBufs.end, fetchBufs.disc);
        return {off: fetchBufs.off, idx: fetchBufs.idx, start: fetchBufs.start, end: fetchBufs.end, disc: fetchBufs.disc, total};
      }catch(err){
        if(err.code !== -5){ throw err; }
        cap *= 4;                                    // the feeds outgrew the buffers: larger ones, once more
      }
    }
  }

  function fetchRows(idx){
    const m = idx.length;
    const s = new BigInt64Array(m), e = new BigInt64Array(m), u = new Int32Array(m), d = new Int32Array(m);
    if(m > 0){
      native.fetchRows(ctx, idx, m, s, e, u, d);
    }
    return {start: s, end: e, user: u, disc: d};
  }

  // ---- persistence (SURVEY.md 8f-4; the reference keeps sessions in memory only, sessionStore.js:6) -----------------
  // save(dir): the four column files of pie_save_columns + tokens.json {userIds, tokenHash per row (null = gone),
  // lastPurge}.  Tokens themselves are never stored, only their sha256 (as in the reference's Map keys, :8-10).
  function save(dir){
    noBase('save');
    flush();
    native.saveColumns(ctx, dir);
    const doc = {format: 'pie-tokens', version: 1, rows: rows.length, userIds, tokenHash: rows.map(r => r.tokenHash), lastPurge};
    fs.writeFileSync(path.join(dir, 'tokens.json'), JSON.stringify(doc));
  }
  function restore(dir){
    if(rows.length !== 0){ throw new Error('restore() needs an empty store'); }
    const doc = JSON.parse(fs.readFileSync(path.join(dir, 'tokens.json'), 'utf8'));
    if(doc.format !== 'pie-tokens' || doc.version !== 1){ throw new Error('not a pie-tokens file'); }
    const loaded = native.loadColumnsDir(ctx, dir);
    if(loaded.rows !== doc.rows || doc.tokenHash.length !== doc.rows || !Array.isArray(doc.userIds) || doc.userIds.length < loaded.users && loaded.rows > 0){
      throw new Error('tokens.json does not match the column files');
    }
    const n = doc.rows;
    const s = new BigInt64Array(n), e = new BigInt64Array(n), u = new Int32Array(n), d = new Int32Array(n);
    if(n > 0){ native.readColumns(ctx, s, e, u, d); }
    doc.userIds.forEach(id => denseUser(id));
    for(let i = 0; i < n; i++){
      const dead = e[i] === END_NONE;
      const lost = dead || doc.tokenHash[i] === null;
      rows.push({tokenHash: lost ? null : doc.tokenHash[i], userId: userIds[u[i]], user: u[i], disc: d[i],
        createdAt: Number(s[i]), expiresAt: dead ? 0 : Number(e[i])});
      if(dead){ rows[i].gone = true; nGone++; }
      if(!lost){ rowOfToken.set(doc.tokenHash[i], i); }
    }
    uploaded = n;
    lastPurge = doc.lastPurge;
    generation++;
  }

  function close(){
    native.ctxDestroy(ctx);
  }

  return {
    createSession, getSession, touchSession, deleteSession, deleteSessionsForUser, purgeExpiredSessions, purgeRetention,
    SESSION_TTL_MS, SESSION_COOKIE_NAME,
    scanFeeds, scanDevice, userFeed, scanBatchDevice, batchUserFeed, batchFetch, BATCH_MAX, fetchRows, expiredRows, archivedRows, flush, close, save, restore,
    compact, compactions: () => compactions, tableRows: () => base + rows.length, baseRows: () => base,
    userIds: () => userIds,
    userIndexOf: userId => (userIndex.has(userId) ? userIndex.get(userId) : -1),
    size: () => rowOfToken.size,
    generation: () => generation,
    native, ctx
  };
}

// module-level singleton with the reference's export names; created on first use so that requiring this file
// on a machine without a GPU fails at the first call, with the library's own error text.
let shared = null;
function store(){
  if(shared === null){
    shared = createStore();
  }
  return shared;
}

module.exports = {
  createSession: (userId, disciplineId) => store().createSession(userId, disciplineId),
  getSession: token => store().getSession(token),
  touchSession: token => store().touchSession(token),
  deleteSession: token => store().deleteSession(token),
  deleteSessionsForUser: userId => store().deleteSessionsForUser(userId),
  purgeExpiredSessions: () => store().purgeExpiredSessions(),
  SESSION_TTL_MS,
  SESSION_COOKIE_NAME,
  createStore,
  sharedStore: store
};
