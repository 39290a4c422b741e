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

// ---- table / CSV / message forms of a payload (SURVEY.md 8f-3; /root/reference/server/webhookDispatcher.js:276-342) ------
// The reference exports every dispatched record three ways: a row object keyed by EXPORT_COLUMNS (buildTableRow :276-305),
// the same object with null / undefined blanked (buildMessagePayload :307-313), and one CSV line in column order
// (buildCsvRow :340-342 over csvEscape :332-338).  The column set here is session-derived ([DERIVED]: the reference's 24
// columns describe show entries, which this path does not hold); the builders' rules are the reference's:
//   table row   columns.map(c => row[c] ?? '')                               (:436, :567)
//   message     every column present, undefined / null -> ''                 (:308-312)
//   csvEscape   null / undefined -> ''; String(value); quoted iff it holds '"', ',' , '\n' or '\r', inner '"' doubled (:332-338)
//   csv row     columns.map(c => csvEscape(row[c] ?? '')).join(',')          (:341)
// Parity unpinned: webhookDispatcher.js does not load on this image's Node 12 (`?.` / `??`, axios absent); the rules are
// checked against vectors hand-derived from the cited lines and the self-consistency assertions of
// /root/reference/scripts/simulate-webhook.js:71-95 (row follows column order, message mirrors the row, header = columns).
const EXPORT_COLUMNS = ['sessionRow', 'userId', 'discipline', 'createdAt', 'expiredAt'];

const blank = v => (v === undefined || v === null ? '' : v);

function buildTableRow(payload){
  const p = payload || {};
  return {
    sessionRow: p.sessionRow === null || p.sessionRow === undefined ? '' : p.sessionRow,   // numeric like delaySec (:299)
    userId: p.userId || '',
    discipline: p.discipline || '',
    createdAt: p.createdAt || '',
    expiredAt: p.expiredAt || ''
  };
}

function buildMessagePayload(rowObject){
  const row = rowObject || {};
  return EXPORT_COLUMNS.reduce((acc, column) => {
    acc[column] = blank(row[column]);
    return acc;
  }, {});
}

function csvEscape(value){
  const str = value === null || value === undefined ? '' : String(value);
  if(str.indexOf('"') >= 0 || str.indexOf(',') >= 0 || /[\n\r]/.test(str)){
    return '"' + str.replace(/"/g, '""') + '"';
  }
  return str;
}

function buildCsvRow(rowObject){
  return EXPORT_COLUMNS.map(column => csvEscape(blank(rowObject[column]))).join(',');
}

// per-entry payload, the shape of dispatchEntryEvent (:425-455): event, schemaVersion 2, table {columns, row},
// csv {header, row}, message, plus the record itself
function buildEntryPayload(event, sessionPayload, dispatchedAt){
  const rowObject = buildTableRow(sessionPayload);
  return {
    event,
    schemaVersion: 2,
    dispatchedAt: dispatchedAt === undefined ? new Date().toISOString() : dispatchedAt,
    table: {columns: EXPORT_COLUMNS, row: EXPORT_COLUMNS.map(column => blank(rowObject[column]))},
    csv: {header: EXPORT_COLUMNS, row: buildCsvRow(rowObject)},
    message: buildMessagePayload(rowObject),
    session: sessionPayload
  };
}

// whole-queue payload, the shape of dispatchShowEvent (:556-577): table.rows, csv.rows, message.entries
function buildQueuePayload(event, sessionPayloads, dispatchedAt){
  const tableRows = sessionPayloads.map(buildTableRow);
  return {
    event,
    schemaVersion: 2,
    dispatchedAt: dispatchedAt === undefined ? new Date().toISOString() : dispatchedAt,
    table: {columns: EXPORT_COLUMNS, rows: tableRows.map(row => EXPORT_COLUMNS.map(column => blank(row[column])))},
    csv: {header: EXPORT_COLUMNS, rows: tableRows.map(row => buildCsvRow(row))},
    message: {entries: tableRows},
    entries: sessionPayloads
  };
}

// CSV text of a queue straight from the fetched columns: header line + one line per row, '\n' between lines — the bytes
// of [EXPORT_COLUMNS.join(',')].concat(rows.map(buildCsvRow)).join('\n').  Native writer when the addon has it.
function queueCsv(store, rows){
  const cols = store.fetchRows(rows);
  const discIds = disciplineConfig.DISCIPLINES.map(d => d.id);
  if(store.native && typeof store.native.serializeCsv === 'function'){
    const buf = store.native.serializeCsv(rows, rows.length, cols.start, cols.end, cols.user, cols.disc, store.userIds(), discIds, EXPORT_COLUMNS.join(','));
    if(buf !== null){ return buf; }
  }
  const lines = [EXPORT_COLUMNS.join(',')];
  for(let i = 0; i < rows.length; i++){
    lines.push(buildCsvRow(buildTableRow(buildExpiredSessionPayload(rows[i], cols, i, store.userIds()))));
  }
  return Buffer.from(lines.join('\n'), 'utf8');
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

module.exports = {dispatchExpiredSessions, dispatchArchivedGroups, buildExpiredSessionPayload, EXPORT_COLUMNS, buildTableRow,
  buildMessagePayload, csvEscape, buildCsvRow, buildEntryPayload, buildQueuePayload, queueCsv};
