'use strict';
// Batched feed aggregation: ONE device scan produces the feeds of every user; a request reads its slice.
// Replaces the reference's per-request chain syncCalendarEvents -> listCalendarEvents
// (/root/reference/server/storage/sqlProvider.js:274-298) for the session-derived events of this build:
// the window predicate (:284), the ordering (:276) and the per-user grouping run on the GPU; the host only
// serialises the selected rows into the event objects of calendarFeed.js:66-79.
const calendarFeed = require('./calendarFeed');
const disciplineConfig = require('./disciplineConfig');

function createFeedService(store, options){
  const opts = options || {};
  const monthsBack = opts.monthsBack === undefined ? 2 : opts.monthsBack;

  // Requests that arrive in the same millisecond, with the same window and discipline filter and no change to the
  // store in between, see exactly the same table at exactly the same `now`: they share ONE device scan (each still
  // reads its own slice).  Nothing is ever served from a scan taken at another instant or another state of the store.
  let last = null;        // the device holds the result of this scan; .full = host copy of every user's feed, if taken
  let scansRun = 0;
  function key(query){
    const q = query || {};
    const now = q.now === undefined ? Date.now() : q.now;
    const cutoff = q.cutoff === undefined ? calendarFeed.getCalendarCutoffTimestamp(monthsBack, now) : q.cutoff;
    const filter = q.disciplines === undefined ? null : JSON.stringify(q.disciplines);
    const gen = typeof store.generation === 'function' ? store.generation() : null;
    return {gen, now, cutoff, filter, disciplines: q.disciplines};
  }
  const current = k => last !== null && k.gen !== null && last.gen === k.gen && last.now === k.now && last.cutoff === k.cutoff && last.filter === k.filter;

  // whole result on the host (every user's feed): allFeeds and callers that want counts / offsets
  function scan(query){
    const k = key(query);
    if(current(k) && last.full){ return last.full; }
    const res = store.scanFeeds({now: k.now, cutoff: k.cutoff, disciplines: k.disciplines});
    scansRun++;
    // scanFeeds hands out views of buffers it reuses: the shared result stays valid because any later scan replaces it
    last = {gen: k.gen, now: k.now, cutoff: k.cutoff, filter: k.filter, full: {now: k.now, cutoff: k.cutoff, res}};
    return last.full;
  }

  // one user's rows: the scan stays in HBM, the request reads its slice (two small copies)
  function userRows(u, query){
    const k = key(query);
    if(!current(k)){
      store.scanDevice({now: k.now, cutoff: k.cutoff, disciplines: k.disciplines});
      scansRun++;
      last = {gen: k.gen, now: k.now, cutoff: k.cutoff, filter: k.filter, full: null};
    }
    return store.userFeed(u);
  }

  // per-discipline constants of the event object, computed once with the JS mirror of parseCalendarMetadata; a
  // discipline whose name would make the title parse differently per row (a '#<digits>' of its own) disables the
  // native serialiser
  let perDisc = null;
  function nativeTable(){
    if(perDisc !== null){ return perDisc; }
    const table = [];
    for(const d of disciplineConfig.DISCIPLINES){
      const probe = calendarFeed.eventFromRow(123456789, 0n, 0n, d.name);
      if(probe.showNumber !== 123456789){ perDisc = false; return perDisc; }
      const nameJson = JSON.stringify(d.name);
      table.push([nameJson.slice(1, nameJson.length - 1), JSON.stringify(probe.eventName), JSON.stringify(probe.color)]);
    }
    perDisc = table;
    return perDisc;
  }

  // {"events":[...]} as bytes for one slice, by the native serialiser when it covers the rows, else via JSON.stringify
  function eventsJsonFromRows(idx){
    const cols = store.fetchRows(idx);
    const table = nativeTable();
    if(table && store.native && typeof store.native.serializeEvents === 'function'){
      const buf = store.native.serializeEvents(idx, idx.length, cols.start, cols.end, cols.disc, table);
      if(buf !== null){ return buf; }
    }
    const events = [];
    for(let i = 0; i < idx.length; i++){
      const d = disciplineConfig.DISCIPLINES[cols.disc[i]];
      events.push(calendarFeed.eventFromRow(idx[i], cols.start[i], cols.end[i], d ? d.name : 'Session'));
    }
    return Buffer.from(JSON.stringify({events}));
  }

  function eventsJsonFromColumns(idx, start, end, disc){
    const table = nativeTable();
    if(table && store.native && typeof store.native.serializeEvents === 'function'){
      const buf = store.native.serializeEvents(idx, idx.length, start, end, disc, table);
      if(buf !== null){ return buf; }
    }
    const events = [];
    for(let i = 0; i < idx.length; i++){
      const d = disciplineConfig.DISCIPLINES[disc[i]];
      events.push(calendarFeed.eventFromRow(idx[i], start[i], end[i], d ? d.name : 'Session'));
    }
    return Buffer.from(JSON.stringify({events}));
  }

  // the same slice as iCalendar text (new functionality, see calendarFeed.toICalendar): native writer when it covers
  // the rows, else the JS emitter over the event objects
  function icsFromRows(idx, dtstamp){
    const cols = store.fetchRows(idx);
    const stamp = dtstamp === undefined ? Date.now() : dtstamp;
    if(store.native && typeof store.native.serializeICal === 'function'){
      const summaries = disciplineConfig.DISCIPLINES.map(d => calendarFeed.icsEscape(d.name + ' session #'));
      const buf = store.native.serializeICal(idx, idx.length, cols.start, cols.end, cols.disc, summaries, stamp);
      if(buf !== null){ return buf; }
    }
    const events = [];
    for(let i = 0; i < idx.length; i++){
      const d = disciplineConfig.DISCIPLINES[cols.disc[i]];
      events.push(calendarFeed.eventFromRow(idx[i], cols.start[i], cols.end[i], d ? d.name : 'Session'));
    }
    return Buffer.from(calendarFeed.toICalendar(events, {dtstamp: stamp}), 'utf8');
  }

  function icsForUser(userId, query, dtstamp){
    const u = store.userIndexOf(userId);
    if(u < 0){ return Buffer.from(calendarFeed.toICalendar([], {dtstamp: dtstamp === undefined ? Date.now() : dtstamp}), 'utf8'); }
    return icsFromRows(userRows(u, query), dtstamp);
  }

  // response body bytes for one user's feed
  function eventsJsonForUser(userId, query){
    const u = store.userIndexOf(userId);
    if(u < 0){ return Buffer.from('{"events":[]}'); }
    return eventsJsonFromRows(userRows(u, query));
  }

  function eventsFromRows(idx){
    const cols = store.fetchRows(idx);
    const events = [];
    for(let i = 0; i < idx.length; i++){
      const d = disciplineConfig.DISCIPLINES[cols.disc[i]];
      events.push(calendarFeed.eventFromRow(idx[i], cols.start[i], cols.end[i], d ? d.name : 'Session'));
    }
    return events;          // already (startTs asc, row asc): the device ordered the bucket
  }

  // events of one user
  function eventsForUser(userId, query){
    const u = store.userIndexOf(userId);
    if(u < 0){
      return [];
    }
    return eventsFromRows(userRows(u, query));
  }

  // feeds of every user from ONE scan: Map userId -> events
  function allFeeds(query){
    const {res} = scan(query);
    const feeds = new Map();
    for(let u = 0; u < res.userIds.length; u++){
      const lo = Number(res.offsets[u]), hi = Number(res.offsets[u + 1]);
      if(hi > lo){
        feeds.set(res.userIds[u], eventsFromRows(res.idx.subarray(lo, hi)));
      }
    }
    return feeds;
  }

  // ---- the requests of one event-loop turn, ONE table pass ---------------------------------------------------------
  // requests: [{userId, query}] -> response bodies ({"events":[...]} bytes), in request order.  Requests are grouped by
  // their (now, cutoff, discipline filter); up to store.BATCH_MAX distinct groups share one batched device scan
  // (pie_scan_batch: the key column is streamed once, every candidate row is evaluated against all the queries), each
  // request then reads its user's slice of its query's result.  Same bytes as eventsJsonForUser, request by request.
  let batchesRun = 0;
  const timing = {scanNs: 0, fetchNs: 0, serializeNs: 0};   // host wall time spent in the three native stages of the batches
  function eventsJsonForRequests(requests){
    const bodies = new Array(requests.length);
    const groups = [];                 // {k, members: [request index]}
    const byKey = new Map();
    requests.forEach((r, i) => {
      const k = key(r.query);
      const id = k.now + '|' + k.cutoff + '|' + k.filter;
      let g = byKey.get(id);
      if(g === undefined){
        g = {k, members: []};
        byKey.set(id, g);
        groups.push(g);
      }
      g.members.push(i);
    });
    const maxQ = store.BATCH_MAX || 16;
    for(let at = 0; at < groups.length; at += maxQ){
      const chunk = groups.slice(at, at + maxQ);
      const ts = process.hrtime.bigint();
      store.scanBatchDevice(chunk.map(g => ({now: g.k.now, cutoff: g.k.cutoff, disciplines: g.k.disciplines})));
      timing.scanNs += Number(process.hrtime.bigint() - ts);
      batchesRun++;
      last = null;                     // the device now holds a batch result, not a single scan's
      // every request's rows and their columns in ONE native call (pie_batch_fetch_requests), then one serialiser call each
      const members = [];
      chunk.forEach((g, qi) => { for(const i of g.members){ members.push([i, qi, store.userIndexOf(requests[i].userId)]); } });
      if(typeof store.batchFetch === 'function' && members.length > 1){
        const qis = Int32Array.from(members, m => m[1]), us = Int32Array.from(members, m => m[2]);
        const t0 = process.hrtime.bigint();
        const f = store.batchFetch(qis, us);
        const t1 = process.hrtime.bigint();
        members.forEach((m, k) => {
          const a = Number(f.off[k]), b = Number(f.off[k + 1]);
          bodies[m[0]] = m[2] < 0 || b === a ? Buffer.from('{"events":[]}')
            : eventsJsonFromColumns(f.idx.subarray(a, b), f.start.subarray(a, b), f.end.subarray(a, b), f.disc.subarray(a, b));
        });
        timing.fetchNs += Number(t1 - t0);
        timing.serializeNs += Number(process.hrtime.bigint() - t1);
      }else{
        for(const m of members){
          bodies[m[0]] = m[2] < 0 ? Buffer.from('{"events":[]}') : eventsJsonFromRows(store.batchUserFeed(m[1], m[2]));
        }
      }
    }
    return bodies;
  }

  return {scan, eventsForUser, eventsJsonForUser, eventsJsonForRequests, icsForUser, allFeeds, scansRun: () => scansRun,
    batchesRun: () => batchesRun, timing: () => Object.assign({}, timing)};
}

module.exports = {createFeedService};
