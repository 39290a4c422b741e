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

// ---- iCalendar text (RFC 5545) for a list of events.  NEW functionality: the reference only CONSUMES .ics
// (node-ical, calendarFeed.js:46); there is no producer to mirror.  The fields are chosen so that the reference's
// own consumer loop (calendarFeed.js:52-80) maps each VEVENT back to the event it came from: UID -> id,
// SUMMARY -> title, DTSTART / DTEND (UTC date-times) -> start / end, DESCRIPTION / LOCATION when non-empty.
// iCalendar date-times carry whole seconds, so startTs / endTs come back floored to the second.
const ICS_PRODID = '-//sph-pie_amd//session feed//EN';

function icsEscape(text){
  return String(text).replace(/\\/g, '\\\\').replace(/;/g, '\\;').replace(/,/g, '\\,').replace(/\r\n|\n|\r/g, '\\n');
}

// content lines longer than 75 octets are folded: CRLF + one space (RFC 5545 3.1); never inside a UTF-8 sequence
function icsFold(line){
  const bytes = Buffer.from(line, 'utf8');
  if(bytes.length <= 75){ return line; }
  const parts = [];
  let at = 0, room = 75;
  while(at < bytes.length){
    let end = Math.min(at + room, bytes.length);
    while(end < bytes.length && (bytes[end] & 0xC0) === 0x80){ end--; }
    parts.push(bytes.slice(at, end).toString('utf8'));
    at = end;
    room = 74;                                       // continuation lines start with a space
  }
  return parts.join('\r\n ');
}

function icsUtc(ms){
  const iso = new Date(Math.floor(ms / 1000) * 1000).toISOString();     // YYYY-MM-DDTHH:mm:ss.000Z
  return iso.slice(0, 4) + iso.slice(5, 7) + iso.slice(8, 10) + 'T' + iso.slice(11, 13) + iso.slice(14, 16) + iso.slice(17, 19) + 'Z';
}

// events: objects of the calendarFeed.js:66-79 shape.  options.dtstamp (ms) fixes DTSTAMP (default: now).
function toICalendar(events, options){
  const stamp = icsUtc(options && options.dtstamp !== undefined ? options.dtstamp : Date.now());
  const lines = ['BEGIN:VCALENDAR', 'VERSION:2.0', 'PRODID:' + ICS_PRODID, 'CALSCALE:GREGORIAN'];
  for(const ev of (Array.isArray(events) ? events : [])){
    if(!ev || !Number.isFinite(ev.startTs)){ continue; }
    lines.push('BEGIN:VEVENT');
    lines.push(icsFold('UID:' + icsEscape(ev.id)));
    lines.push('DTSTAMP:' + stamp);
    lines.push('DTSTART:' + icsUtc(ev.startTs));
    if(ev.endTs !== null && ev.endTs !== undefined && Number.isFinite(ev.endTs)){ lines.push('DTEND:' + icsUtc(ev.endTs)); }
    lines.push(icsFold('SUMMARY:' + icsEscape(ev.title)));
    if(ev.description){ lines.push(icsFold('DESCRIPTION:' + icsEscape(ev.description))); }
    if(ev.location){ lines.push(icsFold('LOCATION:' + icsEscape(ev.location))); }
    lines.push('END:VEVENT');
  }
  lines.push('END:VCALENDAR');
  return lines.join('\r\n') + '\r\n';
}

module.exports = {fetchCalendarFeed, getCalendarCutoffTimestamp, parseCalendarMetadata, eventFromRow, windowAndDedup, orderEvents, END_NONE,
  toICalendar, icsEscape, icsFold, icsUtc, ICS_PRODID};
