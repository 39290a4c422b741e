#!/usr/bin/env node
// TEST INFRASTRUCTURE — golden vectors for the window scalar of /root/reference/server/calendarFeed.js:33-38
// (getCalendarCutoffTimestamp: local midnight of `now`, minus monthsBack calendar months with JS overflow rules).  That
// module cannot be imported here (it needs node-ical and Node >= 14), but the function is exactly three JS `Date` calls —
// `now.setHours(0, 0, 0, 0); now.setMonth(now.getMonth() - monthsBack); now.getTime()` — on a LOCAL-time Date, so the
// vectors below are produced by the JS engine's own Date implementation under several real time zones (DST included), the
// semantics the reference relies on.  Data only; no reference source is copied.
// usage: node oracle/gen_cutoff_golden.js > tests/golden/cutoff_zones.json
'use strict';
const {execFileSync} = require('child_process');
const ZONES = ['UTC', 'America/New_York', 'Europe/Berlin', 'Asia/Kolkata', 'Australia/Lord_Howe', 'America/Havana', 'Pacific/Apia'];

function cutoff(nowMs, monthsBack){
  const now = new Date(nowMs);
  now.setHours(0, 0, 0, 0);
  now.setMonth(now.getMonth() - monthsBack);
  return now.getTime();
}

if(process.argv[2] === '--zone'){
  const cases = [];
  let s = 0x5EED5EED >>> 0;
  const rnd = () => { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; };
  const push = (now, back) => cases.push([now, back, cutoff(now, back)]);
  // month-end overflow (Apr 30 - 2 -> "Feb 30" -> Mar 2), leap years, year wrap, DST change days of either hemisphere
  for(const iso of ['2025-04-30T12:00:00Z', '2024-04-30T12:00:00Z', '2025-12-31T23:30:00Z', '2025-05-31T05:00:00Z', '2025-03-31T00:00:00Z',
    '2025-01-31T18:00:00Z', '2024-02-29T10:00:00Z', '2025-03-09T12:00:00Z', '2025-03-30T01:30:00Z', '2025-11-02T06:30:00Z', '2025-10-26T00:30:00Z',
    '2025-04-06T03:00:00Z', '2025-03-09T05:30:00Z', '2019-03-10T05:30:00Z', '2025-10-05T15:00:00Z', '2011-12-31T12:00:00Z', '1970-01-01T00:00:00Z', '2000-03-01T00:00:00Z']){
    for(const back of [2, 1, 0, 3, 12, 14, -1, -2, 25]){ push(Date.parse(iso), back); }
  }
  for(let i = 0; i < 300; i++){ push(1500000000000 + Math.floor(rnd() * 4.0e11), 2); }                       // 2017 .. 2030, the default
  for(let i = 0; i < 200; i++){ push(Math.floor(rnd() * 4.1e12), Math.floor(rnd() * 40) - 8); }              // 1970 .. 2099
  process.stdout.write(JSON.stringify(cases));
}else{
  const zones = {};
  for(const tz of ZONES){
    zones[tz] = JSON.parse(execFileSync(process.execPath, [__filename, '--zone'], {env: Object.assign({}, process.env, {TZ: tz}), maxBuffer: 1 << 26}).toString());
  }
  console.log(JSON.stringify({provenance: 'JS engine Date arithmetic (Node ' + process.version + '), one process per TZ: now=new Date(nowMs); now.setHours(0,0,0,0); ' +
    'now.setMonth(now.getMonth()-monthsBack); now.getTime() — the three calls of calendarFeed.js:34-37; rows are [nowMs, monthsBack, cutoffMs]', zones}));
}
