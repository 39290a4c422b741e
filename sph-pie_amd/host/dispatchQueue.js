'use strict';
// "next" row of SURVEY.md §8f-1 on the host side: drain the device's ordered queue of newly-expired sessions
// through a dispatcher, one payload at a time, in queue order.  Control flow and summary shape follow the
// reference's per-entry archive dispatch (/root/reference/server/webhookDispatcher.js:519-553: sequential
// `await` per entry, per-entry results, {success, dispatched, failed, total, results[, error]}); the payload
// fields are session-derived ([DERIVED]: the reference dispatches show entries, not sessions).
// The transport (`send`) is the caller's: the reference's axios POST / handshake machinery is out of scope.
const disciplineConfig = require('./disciplineConfig');
const {END_NONE} = require('./calendarFeed');

function buildExpiredSessionPayload(row, cols, i, userIds){
  const d = disciplineConfig.DISCIPLINES[cols.disc[i]];
  const endTs = cols.end[i] === END_NONE ? null : Number(cols.end[i]);
  return {
    sessionRow: row,
    userId: userIds[cols.user[i]] === undefined ? '' : userIds[cols.user[i]],
    discipline: d ? d.id : '',
    createdAt: new Date(Number(cols.start[i])).toISOString(),
    expiredAt: endTs === null ? '' : new Date(endTs).toISOString()
  };
}

// store: a device-backed session store (host/sessionStore.js createStore()).  send(payload, meta) -> Promise of
// {success: boolean, ...}.  Rows with prevNow < expiresAt <= now are dispatched in ascending row order.
async function dispatchExpiredSessions(store, prevNow, now, send){
  return drain(store, store.expiredRows(prevNow, now), send, 'session.expired', 'session-expired-entry');
}

// The reference's daily-archive trigger (/root/reference/server/storage/sqlProvider.js:758-861) on sessions: every
// session of every user whose earliest session is at least windowMs (default 12 h = AUTO_ARCHIVE_WINDOW_MS, :9) old,
// users in first-appearance order, dispatched one at a time.
async function dispatchArchivedGroups(store, now, send, windowMs){
  return drain(store, store.archivedRows(now, windowMs), send, 'session.archived', 'session-archive-entry');
}

async function drain(store, rows, send, event, kind){
  if(rows.length === 0){
    return {success: true, dispatched: 0, failed: 0, total: 0, results: []};
  }
  const cols = store.fetchRows(rows);
  const results = [];
  for(let i = 0; i < rows.length; i++){                       // sequential await: order is observable
    const payload = buildExpiredSessionPayload(rows[i], cols, i, store.userIds());
    let res;
    try{
      res = await send(payload, {event, kind, sessionRow: rows[i]});
    }catch(err){
      res = {success: false, error: err && err.message ? err.message : String(err)};
    }
    results.push(Object.assign({}, res, {sessionRow: rows[i]}));
  }
  const failures = results.filter(r => r && r.success === false);
  const summary = {
    success: failures.length === 0,
    dispatched: results.filter(r => !r || r.success !== false).length,
    failed: failures.length,
    total: rows.length,
    results
  };
  if(failures.length){
    summary.error = event === 'session.expired' ? 'One or more expired-session payloads failed to dispatch'
      : 'One or more archived-session payloads failed to dispatch';
  }
  return summary;
}

module.exports = {dispatchExpiredSessions, dispatchArchivedGroups, buildExpiredSessionPayload};
