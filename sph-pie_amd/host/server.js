'use strict';
// The HTTP seam that is kept: GET /api/calendar -> 200 {"events":[...]} sorted by startTs ascending, with the
// reference's auth conventions (/root/reference/server/index.js): cookie mt_session (:594-610), 401
// {"error":"Authentication required"} (:656-658), 403 {"error":"Insufficient permissions"} (:666-668), 423
// {"error":"Password reset required"} (:99-111), admin bypass (:661-664), allowed roles drones.lead/operator/crew
// (:45-50,293), thrown errors -> 500 {"error":"Internal server error","detail":msg} (:526-536).
// Node core http only (no Express here: nothing can be installed).  Users come from a caller-supplied
// findUserById — the user directory / password store is out of scope.
const http = require('http');
const disciplineConfig = require('./disciplineConfig');
const {createFeedService} = require('./feedService');

const lowerTrim = v => (typeof v === 'string' ? v.trim().toLowerCase() : '');

function readSessionToken(req, cookieName){
  const header = req.headers ? req.headers.cookie : undefined;
  if(!header){
    return null;
  }
  for(const piece of header.split(';')){
    const c = piece.trim();
    if(c && c.indexOf(cookieName + '=') === 0){
      return decodeURIComponent(c.slice(cookieName.length + 1));
    }
  }
  return null;
}

function sendJson(res, status, body){
  const text = JSON.stringify(body);
  res.writeHead(status, {'Content-Type': 'application/json; charset=utf-8', 'Content-Length': Buffer.byteLength(text)});
  res.end(text);
}

function createServer(options){
  const store = options.store;
  const findUserById = options.findUserById;
  const feeds = options.feeds || createFeedService(store, options);
  const drone = disciplineConfig.findDiscipline('drones') || disciplineConfig.DEFAULT_DISCIPLINE;
  const readRoles = new Set(['lead', 'operator', 'crew'].map(lv => (drone ? disciplineConfig.getRoleKey(drone.id, lv) : null)).filter(Boolean));

  function authenticate(req){
    const token = readSessionToken(req, store.SESSION_COOKIE_NAME);
    if(!token){
      return null;
    }
    const session = store.getSession(token);
    if(!session){
      return null;
    }
    const user = findUserById(session.userId);
    if(!user){
      store.deleteSession(token);                       // index.js:89-92
      return null;
    }
    return user;
  }

  // Request coalescing (options.coalesce): "calendarFeed per-request loop -> batched GPU scan".  Feed requests that arrive
  // in one turn of the event loop are answered together: their (now, cutoff, filter) queries go to the device as ONE
  // batched scan, and every response is written from its own slice.  A failure answers every request of the batch with
  // the 500 body of /root/reference/server/index.js:526-536.
  const coalesce = options.coalesce === true;
  let pending = [];
  function flushPending(){
    const batch = pending;
    pending = [];
    try{
      const bodies = feeds.eventsJsonForRequests(batch.map(p => ({userId: p.userId, query: p.query})));
      batch.forEach((p, i) => {
        p.res.writeHead(200, {'Content-Type': 'application/json; charset=utf-8', 'Content-Length': bodies[i].length});
        p.res.end(bodies[i]);
      });
    }catch(err){
      console.error(err);
      for(const p of batch){
        if(!p.res.headersSent){
          sendJson(p.res, 500, {error: 'Internal server error', detail: err && err.message ? err.message : String(err)});
        }
      }
    }
  }

  function handleCalendar(req, res, asIcs){
    const user = authenticate(req);
    if(user && user.needsPasswordReset){
      return sendJson(res, 423, {error: 'Password reset required'});
    }
    if(!user){
      return sendJson(res, 401, {error: 'Authentication required'});
    }
    const roles = Array.isArray(user.roles) ? user.roles : [];
    if(roles.indexOf('admin') < 0 && !roles.some(r => readRoles.has(lowerTrim(r)))){
      return sendJson(res, 403, {error: 'Insufficient permissions'});
    }
    const query = options.query ? options.query(req, user) : undefined;
    if(asIcs){
      // NEW route (the reference has no .ics producer): the same feed, same auth, as iCalendar text
      const body = feeds.icsForUser(user.id, query);
      res.writeHead(200, {'Content-Type': 'text/calendar; charset=utf-8', 'Content-Length': body.length});
      return res.end(body);
    }
    if(coalesce && typeof feeds.eventsJsonForRequests === 'function'){
      // batched: the request joins the batch of this event-loop turn; one device scan answers them all
      pending.push({res, userId: user.id, query});
      if(pending.length === 1){
        setImmediate(flushPending);
      }
      return undefined;
    }
    if(typeof feeds.eventsJsonForUser === 'function'){
      // body bytes straight from the native serialiser (same bytes as JSON.stringify({events}))
      const body = feeds.eventsJsonForUser(user.id, query);
      res.writeHead(200, {'Content-Type': 'application/json; charset=utf-8', 'Content-Length': body.length});
      return res.end(body);
    }
    const events = feeds.eventsForUser(user.id, query);
    return sendJson(res, 200, {events});
  }

  return http.createServer((req, res) => {
    try{
      const path = (req.url || '').split('?')[0];
      if(req.method === 'GET' && path === '/api/calendar'){
        return handleCalendar(req, res, false);
      }
      if(req.method === 'GET' && path === '/api/calendar.ics'){
        return handleCalendar(req, res, true);
      }
      if(req.method === 'GET' && path === '/api/health'){
        // existing keys of /root/reference/server/index.js:132-144 kept (status, storage, storageMeta); the device
        // scan statistics ride along under a new key, nothing existing changes meaning
        const st = store.native.stats(store.ctx);
        return sendJson(res, 200, {
          status: 'ok',
          storage: 'MI355X HBM columns',
          storageMeta: {label: 'MI355X HBM columns', driver: 'libpie_hip', sessions: store.size()},
          scan: {rows: st.rows, users: st.users, lastSelected: st.selected, algBytesPerScan: st.algBytes}
        });
      }
      return sendJson(res, 404, {error: 'Not found'});
    }catch(err){
      console.error(err);
      const status = Number.isInteger(err.status) ? err.status : 500;
      const payload = {error: status === 500 ? 'Internal server error' : (err.message || 'Request failed')};
      if(status === 500 && err.message){
        payload.detail = err.message;
      }
      return sendJson(res, status, payload);
    }
  });
}

module.exports = {createServer, readSessionToken};
