'use strict';
// CPU checks of the Node host mirror (no GPU): run with TZ=UTC.  Vectors: tests/golden/hand_derived_h1_h5.json.
process.env.TZ = 'UTC';
const assert = require('assert');
const path = require('path');
const fs = require('fs');

const golden = JSON.parse(fs.readFileSync(path.join(__dirname, '..', '..', '..', 'tests', 'golden', 'hand_derived_h1_h5.json'), 'utf8'));
const dc = require('../disciplineConfig');
const cf = require('../calendarFeed');
const {readSessionToken} = require('../server');
let checks = 0;
const ok = (cond, msg) => { assert.ok(cond, msg); checks++; };
const eq = (a, b, msg) => { assert.deepStrictEqual(a, b, msg); checks++; };

// ---- H5 discipline table, role keys and normalisation (SURVEY.md 8a rows a7 - a10: disciplineConfig.js:18-37, 39-146).
// PARITY UNPINNED for a9 (parseRoleKey / roleMatches*) and a10 (getRoleKey / listRoleKeys / normalizeRole / getDisplayName):
// /root/reference/server/disciplineConfig.js does not parse on this image's Node 12 (`?.` at :59) and the reference holds
// no fixture for it, so these vectors are hand-derived from the cited lines, never reference output.
const g = golden.disciplines;
eq(dc.DISCIPLINES.map(d => d.id), g.ids);
eq(dc.ROLE_LEVELS, g.roles);
eq(dc.DEFAULT_DISCIPLINE.id, g.default);
eq(dc.DISCIPLINES.filter(d => d.forms).map(d => d.id), g.forms);
for(const [id, want] of g.lookups){
  eq(dc.disciplineIndex(id), want === null ? -1 : want, 'lookup ' + id);
  eq(dc.findDiscipline(id), want === null ? null : dc.DISCIPLINES[want]);
}
eq(dc.findDiscipline(42), null);
for(const [key, want] of g.role_keys){ eq(dc.parseRoleKey(key), want, 'parseRoleKey ' + key); }
for(const [role, want] of g.normalize){ eq(dc.normalizeRole(role), want, 'normalizeRole ' + role); }
eq(dc.normalizeRole(7), null);
eq(dc.listRoleKeys().length, 21);
eq(dc.getRoleKey('Drones', 'LEAD'), 'drones.lead');
eq(dc.getRoleKey('drones', 'boss'), null);
ok(dc.roleMatchesLevel('drones.lead', 'lead') && !dc.roleMatchesLevel('drones.lead', 'crew'));
ok(dc.roleMatchesDiscipline('video.crew', 'video') && !dc.roleMatchesDiscipline('video.crew', 'audio'));
eq(dc.getDisplayName('admin'), 'Admin');
eq(dc.getDisplayName('show-control.operator'), 'Show Control Operator');
eq(dc.getDisplayName('bogus'), 'bogus');
eq(dc.disciplineMask('drones'), 16n);
eq(dc.disciplineMask('*'), 127n);
eq(dc.disciplineMask(['audio', 'nope', ' Broadcast ']), 65n);

// ---- cutoff (a11: calendarFeed.js:33-38) under TZ=UTC, incl. month-overflow cases (hand-derived); the JS-engine-made
// vectors for seven real zones (tests/golden/cutoff_zones.json) are replayed per zone by tests/test_node_host.py
for(const c of golden.cutoff_cases_tz_utc){
  eq(cf.getCalendarCutoffTimestamp(c.months_back, c.now_ms), c.expect_ms, 'cutoff ' + c.now_iso);
  eq(new Date(cf.getCalendarCutoffTimestamp(c.months_back, c.now_ms)).toISOString().replace('.000Z', 'Z'), c.expect_iso);
}
eq(cf.getCalendarCutoffTimestamp(undefined, golden.cutoff_cases_tz_utc[0].now_ms), golden.cutoff_cases_tz_utc[0].expect_ms);
ok(cf.getCalendarCutoffTimestamp() <= Date.now());

// ---- title metadata (a13: calendarFeed.js:15-31, module-private there).  PARITY UNPINNED: calendarFeed.js does not load here
// (`?.` at :19, node-ical absent) and the reference has no fixture; vectors hand-derived from the cited lines.
eq(cf.parseCalendarMetadata('Eagles #12 load-in'), {eventName: 'EAGLES', showNumber: 12, color: '#3b82f6'});
eq(cf.parseCalendarMetadata('woz show 7 rehearsal 9'), {eventName: 'WOZ', showNumber: 7, color: '#22c55e'});
eq(cf.parseCalendarMetadata('Zac Brown Band: Love and Fear # 3'), {eventName: 'ZAC', showNumber: 3, color: '#ef4444'});
eq(cf.parseCalendarMetadata('2024 tour'), {eventName: '', showNumber: 2024, color: ''});
eq(cf.parseCalendarMetadata(''), {eventName: '', showNumber: null, color: ''});
eq(cf.parseCalendarMetadata(), {eventName: '', showNumber: null, color: ''});

// ---- H2 window + H3 de-dup (sqlProvider.js:284-295)
const w = golden.window_cases[0];
const toNum = v => (v === 'NaN' ? NaN : v === 'Infinity' ? Infinity : v);
const evs = w.startTs.map((s, i) => ({id: 'e' + i, startTs: toNum(s)}));
eq(cf.windowAndDedup(evs, w.cutoff).map(e => Number(e.id.slice(1))), w.kept_rows);
const dd = golden.dedup_cases[0];
eq(cf.windowAndDedup(dd.ids.map((id, i) => ({id, startTs: 10, row: i})), 0).map(e => e.row), dd.kept_rows);
eq(cf.windowAndDedup(null, 0), []);

// ---- ordering (sqlProvider.js:276), ties keep input order
eq(cf.orderEvents([{startTs: 5, k: 0}, {startTs: 1, k: 1}, {startTs: 5, k: 2}, {startTs: 1, k: 3}]).map(e => e.k), [1, 3, 0, 2]);

// ---- event object shape (calendarFeed.js:66-79)
const ev = cf.eventFromRow(42, 1700000000000n, 1700043200000n, 'Drones');
eq(Object.keys(ev), ['id', 'title', 'description', 'location', 'start', 'end', 'startTs', 'endTs', 'allDay', 'eventName', 'showNumber', 'color']);
eq(ev.start, '2023-11-14T22:13:20.000Z');
eq([ev.startTs, ev.endTs, ev.allDay, ev.eventName, ev.showNumber, ev.id], [1700000000000, 1700043200000, false, 'DRONES', 42, 'session-42']);
const noEnd = cf.eventFromRow(1, 1699920000000n, cf.END_NONE, 'Audio');
eq([noEnd.end, noEnd.endTs, noEnd.allDay], ['', null, true]);

// ---- iCalendar emitter (new functionality).  Round trip: events -> .ics text -> a minimal RFC 5545 reader (unfold,
//      split, unescape: what node-ical hands the reference) -> the reference's consumer loop restated
//      (/root/reference/server/calendarFeed.js:52-80: id from uid, title from summary, start/end Dates, the allDay
//      rule of :64, metadata of :65) -> the events we started from, with timestamps floored to the second.
function icsRead(text){
  const lines = text.replace(/\r\n[ \t]/g, '').split('\r\n');
  const unesc = v => v.replace(/\\(.)/g, (m, ch) => (ch === 'n' || ch === 'N' ? '\n' : ch));
  const toDate = v => new Date(Date.UTC(+v.slice(0, 4), +v.slice(4, 6) - 1, +v.slice(6, 8), +v.slice(9, 11), +v.slice(11, 13), +v.slice(13, 15)));
  const entries = [];
  let cur = null;
  for(const line of lines){
    if(line === 'BEGIN:VEVENT'){ cur = {type: 'VEVENT'}; continue; }
    if(line === 'END:VEVENT'){ entries.push(cur); cur = null; continue; }
    if(!cur){ continue; }
    const at = line.indexOf(':');
    const key = line.slice(0, at), val = line.slice(at + 1);
    if(key === 'UID'){ cur.uid = unesc(val); }
    else if(key === 'SUMMARY'){ cur.summary = unesc(val); }
    else if(key === 'DESCRIPTION'){ cur.description = unesc(val); }
    else if(key === 'LOCATION'){ cur.location = unesc(val); }
    else if(key === 'DTSTART'){ cur.start = toDate(val); }
    else if(key === 'DTEND'){ cur.end = toDate(val); }
  }
  return entries;
}
function consumeLikeTheReference(entries){
  const events = [];
  for(const entry of entries){
    if(!entry || entry.type !== 'VEVENT'){ continue; }
    const start = entry.start instanceof Date ? entry.start : null;
    const end = entry.end instanceof Date ? entry.end : null;
    if(!start){ continue; }
    const id = typeof entry.uid === 'string' && entry.uid ? entry.uid : (entry.summary || 'event') + '-' + start.getTime();
    const allDay = start.getUTCHours() === 0 && start.getUTCMinutes() === 0 && (!end || end.getUTCHours() === 0);
    const meta = cf.parseCalendarMetadata(typeof entry.summary === 'string' ? entry.summary : '');
    events.push({id, title: typeof entry.summary === 'string' ? entry.summary : 'Untitled event',
      description: typeof entry.description === 'string' ? entry.description : '', location: typeof entry.location === 'string' ? entry.location : '',
      start: start.toISOString(), end: end ? end.toISOString() : '', startTs: start.getTime(), endTs: end ? end.getTime() : null,
      allDay, eventName: meta.eventName, showNumber: meta.showNumber, color: meta.color});
  }
  return events;
}
{
  const src = [
    cf.eventFromRow(5, 1700000000123n, 1700043200999n, 'Show Control'),
    cf.eventFromRow(77, 1699920000000n, cf.END_NONE, 'Drones'),                    // midnight UTC, no end: allDay
    cf.eventFromRow(2000000000, -5000000000000n, -4999999999000n, '4D'),           // 1811; title starts with a digit
    {id: 'a;b,c\\d', title: 'Eagles #12, late; "x"\nsecond line ' + 'é'.repeat(70), description: 'desc, with; stuff', location: 'Hall 4',
      startTs: 1700000000000, endTs: null}
  ];
  const text = cf.toICalendar(src, {dtstamp: 1700000000000});
  ok(text.startsWith('BEGIN:VCALENDAR\r\nVERSION:2.0\r\n') && text.endsWith('END:VCALENDAR\r\n'));
  ok(text.split('\r\n').every(l => Buffer.byteLength(l, 'utf8') <= 75), 'folded to 75 octets');
  ok(!/[^\r]\n/.test(text), 'CRLF only');
  const back = consumeLikeTheReference(icsRead(text));
  eq(back.length, src.length);
  const floorS = ms => Math.floor(ms / 1000) * 1000;
  src.forEach((ev, i) => {
    const startTs = floorS(ev.startTs), endTs = ev.endTs === null ? null : floorS(ev.endTs);
    const meta = cf.parseCalendarMetadata(ev.title);
    eq(back[i], {id: ev.id, title: ev.title, description: ev.description || '', location: ev.location || '',
      start: new Date(startTs).toISOString(), end: endTs === null ? '' : new Date(endTs).toISOString(), startTs, endTs,
      allDay: new Date(startTs).getUTCHours() === 0 && new Date(startTs).getUTCMinutes() === 0 && (endTs === null || new Date(endTs).getUTCHours() === 0),
      eventName: meta.eventName, showNumber: meta.showNumber, color: meta.color}, 'ics round trip ' + i);
  });
  eq(cf.toICalendar([], {dtstamp: 0}), 'BEGIN:VCALENDAR\r\nVERSION:2.0\r\nPRODID:' + cf.ICS_PRODID + '\r\nCALSCALE:GREGORIAN\r\nEND:VCALENDAR\r\n');
  eq(cf.toICalendar([{id: 'x', title: 't', startTs: NaN}], {dtstamp: 0}).indexOf('VEVENT'), -1);   // non-finite start: not emitted
}

// ---- table / CSV / message builders (webhookDispatcher.js:276-342; hand-derived vectors — parity unpinned: that module does
// not load on Node 12; the three self-consistency checks are the ones /root/reference/scripts/simulate-webhook.js:71-95 makes)
{
  const dq = require('../dispatchQueue');
  // csvEscape :332-338
  eq(dq.csvEscape(null), ''); eq(dq.csvEscape(undefined), ''); eq(dq.csvEscape(0), '0'); eq(dq.csvEscape(false), 'false');
  eq(dq.csvEscape('plain text'), 'plain text');
  eq(dq.csvEscape('a,b'), '"a,b"');
  eq(dq.csvEscape('say "hi"'), '"say ""hi"""');
  eq(dq.csvEscape('line\nbreak'), '"line\nbreak"');
  eq(dq.csvEscape('cr\rhere'), '"cr\rhere"');
  eq(dq.csvEscape('"'), '""""');
  eq(dq.csvEscape(' semi;colon '), ' semi;colon ');
  const payload = {sessionRow: 0, userId: 'Smith, "Al"', discipline: 'drones', createdAt: '2025-01-02T03:04:05.006Z', expiredAt: ''};
  const row = dq.buildTableRow(payload);
  eq(row, {sessionRow: 0, userId: 'Smith, "Al"', discipline: 'drones', createdAt: '2025-01-02T03:04:05.006Z', expiredAt: ''});   // 0 stays 0 (delaySec rule :299)
  eq(dq.buildTableRow({}), {sessionRow: '', userId: '', discipline: '', createdAt: '', expiredAt: ''});
  eq(dq.buildTableRow(), {sessionRow: '', userId: '', discipline: '', createdAt: '', expiredAt: ''});
  eq(dq.buildCsvRow(row), '0,"Smith, ""Al""",drones,2025-01-02T03:04:05.006Z,');
  eq(dq.buildMessagePayload({sessionRow: 5, userId: null}), {sessionRow: 5, userId: '', discipline: '', createdAt: '', expiredAt: ''});
  const entry = dq.buildEntryPayload('session.expired', payload, '2025-02-02T00:00:00.000Z');
  // simulate-webhook.js:76-95: the table row follows the export column order of the row object, the message mirrors it,
  // the CSV header is the column list
  eq(JSON.stringify(entry.table.row), JSON.stringify(dq.EXPORT_COLUMNS.map(c => (row[c] === undefined || row[c] === null ? '' : row[c]))));
  eq(JSON.stringify(entry.message), JSON.stringify(dq.buildMessagePayload(row)));
  eq(JSON.stringify(entry.csv.header), JSON.stringify(dq.EXPORT_COLUMNS));
  eq([entry.event, entry.schemaVersion, entry.dispatchedAt, entry.csv.row], ['session.expired', 2, '2025-02-02T00:00:00.000Z', dq.buildCsvRow(row)]);
  const queue = dq.buildQueuePayload('session.archived', [payload, {sessionRow: 7, userId: 'u7', discipline: '', createdAt: 'x', expiredAt: 'y'}], 'T');
  eq(queue.table.rows, [[0, 'Smith, "Al"', 'drones', '2025-01-02T03:04:05.006Z', ''], [7, 'u7', '', 'x', 'y']]);
  eq(queue.csv.rows, ['0,"Smith, ""Al""",drones,2025-01-02T03:04:05.006Z,', '7,u7,,x,y']);
  eq(queue.message.entries.length, 2); eq(queue.table.columns, dq.EXPORT_COLUMNS);
}

// ---- fetchCalendarFeed never rejects
(async () => {
  eq(await cf.fetchCalendarFeed(''), []);
  eq(await cf.fetchCalendarFeed(42), []);
  eq(await cf.fetchCalendarFeed('https://example.invalid/feed.ics'), []);
  eq(await cf.fetchCalendarFeed({events: async () => [{id: 'x'}]}), [{id: 'x'}]);
  const quiet = console.error; console.error = () => {};
  eq(await cf.fetchCalendarFeed({events: async () => { throw new Error('boom'); }}), []);
  console.error = quiet;

  // ---- cookie parsing (index.js:594-610)
  eq(readSessionToken({headers: {cookie: 'a=b; mt_session=abc%20d; z=1'}}, 'mt_session'), 'abc d');
  eq(readSessionToken({headers: {cookie: 'xmt_session=1'}}, 'mt_session'), null);
  eq(readSessionToken({headers: {}}, 'mt_session'), null);

  // ---- the addon loads and refuses to work without a GPU (no CPU fallback)
  const pieNative = require('../pieNative');
  if(fs.existsSync(pieNative.ADDON) && fs.existsSync(pieNative.LIB)){
    const native = pieNative.load();
    ok(typeof native.scan === 'function' && typeof native.scanAsync === 'function');
    if(native.deviceCount() === 0){
      assert.throws(() => native.ctxCreate(0), e => e.code === -2 && /no CPU path/.test(e.message));
      assert.throws(() => require('../sessionStore').createSession('u1'), /no CPU path/);
      checks += 2;
    }
  }
  console.log('host cpu_test ok: ' + checks + ' checks');
})().catch(err => { console.error(err); process.exit(1); });
