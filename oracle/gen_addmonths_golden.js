#!/usr/bin/env node
// TEST INFRASTRUCTURE — golden vectors for the retention arithmetic of /root/reference/server/storage/sqlProvider.js
// :991-1009 (_isArchiveExpired / _addMonths).  Those functions cannot be imported here (the module needs sql.js and
// Node >= 14), but their arithmetic is exactly two JS `Date` calls — `date.setMonth(date.getMonth() + months)` on a
// local-time Date — so the vectors below are produced by the JS engine's own Date implementation under TZ=UTC, the
// semantics the reference relies on.  Data only; no reference source is copied.
// usage: TZ=UTC node oracle/gen_addmonths_golden.js > tests/golden/addmonths_utc.json
'use strict';
process.env.TZ = 'UTC';
function addMonths(ts, months){
  const d = new Date(ts);
  if(Number.isNaN(d.getTime())){ return ts; }
  d.setMonth(d.getMonth() + months);
  return d.getTime();
}
const cases = [];
const push = (ts, m) => { const r = addMonths(ts, m); cases.push([ts, m, Number.isNaN(r) ? null : r]); };
let s = 0x5EED5EED >>> 0;
const rnd = () => { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; };
// month-end / leap-year / year-wrap edges
for(const iso of ['2024-12-31T23:59:59.999Z', '2025-12-31T00:00:00.000Z', '2024-02-29T12:00:00.000Z', '2023-12-30T06:00:00.000Z',
  '2025-01-31T01:02:03.004Z', '2025-03-31T00:00:00.000Z', '2025-08-31T12:00:00.000Z', '1970-01-01T00:00:00.000Z',
  '1969-12-31T23:59:59.999Z', '1900-01-31T00:00:00.000Z', '2000-02-29T00:00:00.000Z', '2100-01-31T00:00:00.000Z', '0001-01-01T00:00:00.000Z']){
  for(const m of [2, 1, 0, -1, -2, 12, 14, -14, 25]){ push(Date.parse(iso), m); }
}
// the synthetic corpus range and a wide random range (incl. negative timestamps)
for(let i = 0; i < 400; i++){ push(1700000000000 - Math.floor(rnd() * 10368000000), 2); }
for(let i = 0; i < 400; i++){ push(Math.floor((rnd() - 0.5) * 8e15), Math.floor(rnd() * 50) - 25); }
// range limits of JS Date (+-8.64e15): out-of-range input is returned unchanged, out-of-range result is NaN (null here)
for(const ts of [8640000000000000, 8640000000000001, -8640000000000000, -8640000000000001, 8639999999999999]){ for(const m of [2, -2]){ push(ts, m); } }
console.log(JSON.stringify({provenance: 'JS engine Date arithmetic under TZ=UTC (Node ' + process.version + '): d=new Date(ts); d.setMonth(d.getMonth()+months); d.getTime() — the two calls of sqlProvider.js:999-1009', cases}));
