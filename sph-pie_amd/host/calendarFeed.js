'use strict';
// Host mirror of /root/reference/server/calendarFeed.js (exports :84-87) for the session-scan path.
// The reference CONSUMES a remote .ics through node-ical; this build PRODUCES events from selected session
// rows, so the per-event loop (:52-80) becomes eventFromRow(): same object shape, built from int64 columns.
// Also restated here, because the route needs them on the host: the window + de-dup rule of
// syncCalendarEvents (/root/reference/server/storage/sqlProvider.js:284-295) and the ordering rule of
// listCalendarEvents (:276).

const PALETTE = {woz: '#22c55e', eagles: '#3b82f6', zac: '#ef4444', bsb: '#a855f7', illenium: '#f59e0b'};
const SPECIAL_TITLES = [{needle: 'zac brown band: love and fear', color: '#ef4444', label: 'ZAC'}];

// title -> {eventName, showNumber, color}   (calendarFeed.js:15-31)
function parseCalendarMetadata(summary){
  const text = summary === undefined ? '' : summary;
  const folded = text.toLowerCase();
  let special = null;
  for(const s of SPECIAL_TITLES){
    if(folded.indexOf(s.needle) >= 0){ special = s; break; }
  }
  const word = /^([A-Za-z]+)/.exec(text);
  const eventName = special && special.label ? special.label : (word ? word[1].toUpperCase() : '');
  const hashed = /#\s*(\d+)/.exec(text);
  const bare = hashed ? null : /\b(\d+)\b/.exec(text);
  const showNumber = hashed ? Number(hashed[1]) : (bare ? Number(bare[1]) : null);
  const color = (special && special.color) || PALETTE[eventName.toLowerCase()] || '';
  return {eventName, showNumber, color};
}

// local midnight of `now`, minus monthsBack calendar months with JS overflow rules (calendarFeed.js:33-38).
// nowMs is injectable for tests; the scalar is computed on the host and handed to the device as int64.
function getCalendarCutoffTimestamp(monthsBack, nowMs){
  const back = monthsBack === undefined ? 2 : monthsBack;
  const d = nowMs === undefined ? new Date() : new Date(nowMs);
  d.setHours(0, 0, 0, 0);
  d.setMonth(d.getMonth() - back);
  return d.getTime();
}

const END_NONE = -(2n ** 63n);

// One selected session row -> the event object of calendarFeed.js:66-79 (same keys, same types).
// startTs/endTs are BigInt (int64 columns) or Numbers; endTs sentinel END_NONE maps back to null / ''.
function eventFromRow(row, startTs, endTs, disciplineName){
  const startMs = Number(startTs);
  const hasEnd = endTs !== null && endTs !== undefined && BigInt(endTs) !== END_NONE;
  const endMs = hasEnd ? Number(endTs) : null;
  const start = new Date(startMs);
  const end = hasEnd ? new Date(endMs) : null;
  const title = (disciplineName || 'Session') + ' session #' + row;
  const allDay = start.getUTCHours() === 0 && start.getUTCMinutes() === 0 && (!end || end.getUTCHours() === 0);  // :64
  const meta = parseCalendarMetadata(title);
  return {
    id: 'session-' + row,
    title,
    description: '',
    location: '',
    start: start.toISOString(),
    end: end ? end.toISOString() : '',
    startTs: startMs,
    endTs: endMs,
    allDay,
    eventName: meta.eventName,
    showNumber: meta.showNumber,
    color: meta.color
  };
}

// sqlProvider.js:284-295 — keep finite startTs >= cutoff, skip falsy ids, first occurrence of an id wins
function windowAndDedup(events, cutoff){
  const seen = new Set();
  const kept = [];
  for(const ev of (Array.isArray(events) ? events : [])){
    if(!ev || !Number.isFinite(ev.startTs) || ev.startTs < cutoff){ continue; }
    if(!ev.id || seen.has(ev.id)){ continue; }
    seen.add(ev.id);
    kept.push(ev);
  }
  return kept;
}

// sqlProvider.js:276 ORDER BY start_ts ASC; ties keep input order (stable), which is the a-D row-index rule
function orderEvents(events){
  return events.map((ev, i) => ({ev, i})).sort((a, b) => (a.ev.startTs - b.ev.startTs) || (a.i - b.i)).map(x => x.ev);
}

// The reference fetches feedUrl over HTTPS and never rejects: any failure -> [] (:41-50).  This build has no
// ICS consumer (out of scope: third-party parser + network), so a URL always resolves to []; a feed *source*
// object with an events() method (e.g. feedService) is passed through.
async function fetchCalendarFeed(feedSource){
  if(!feedSource){
    return [];
  }
  if(typeof feedSource === 'string' || typeof feedSource.events !== 'function'){
    return [];
  }
  try{
    const out = await feedSource.events();
    return Array.isArray(out) ? out : [];
  }catch(err){
    console.error('[calendarFeed] Failed to build calendar feed', err);
    return [];
  }
}

module.exports = {fetchCalendarFeed, getCalendarCutoffTimestamp, parseCalendarMetadata, eventFromRow, windowAndDedup, orderEvents, END_NONE};
