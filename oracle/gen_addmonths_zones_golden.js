#!/usr/bin/env node
// TEST INFRASTRUCTURE — golden vectors for the retention arithmetic of /root/reference/server/storage/sqlProvider.js
// :991-1009 (_isArchiveExpired / _addMonths) under REAL time zones.  Those functions cannot be imported here (the module
// needs sql.js and Node >= 14), but their arithmetic is exactly two JS `Date` calls — `date.setMonth(date.getMonth() + months)`
// on a LOCAL-time Date — so the vectors are produced by the JS engine's own Date implementation, one process per zone (the
// zones of gen_cutoff_golden.js: daylight saving in both hemispheres, a 30-minute DST, a change AT midnight, a skipped day).
// Beside the vectors every zone carries the transition table the product's host builds under it
// (sph-pie_amd/host/tzTable.js), which is what the device and the oracle compute with.  Data only.
// usage: node oracle/gen_addmonths_zones_golden.js > tests/golden/addmonths_zones.json
'use strict';
const {execFileSync} = require('child_process');
const path = require('path');
const ZONES = ['UTC', 'America/New_York', 'Europe/Berlin', 'Asia/Kolkata', 'Australia/Lord_Howe', 'America/Havana', 'Pacific/Apia'];

function addMonths(ts, months){
  const d = new Date(ts);
  if(Number.isNaN(d.getTime())){ return ts; }
  d.setMonth(d.getMonth() + months);
  return d.getTime();
}

if(process.argv[2] === '--zone'){
  const {defaultTzTable} = require(path.join(__dirname, '..', 'sph-pie_amd', 'host', 'tzTable.js'));
  const table = defaultTzTable();
  const cases = [];
  let s = 0x5EED5EED >>> 0;
  const rnd = () => { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; };
  const push = (ts, m) => { const r = addMonths(ts, m); cases.push([ts, m, Number.isNaN(r) ? null : r]); };
  // around every transition of 2019 .. 2027: instants whose month shift lands on / next to a change (skipped and repeated
  // local times), and instants next to a change shifted away from it
  for(const tb of table.transitions){
    const t = Number(tb);
    if(t < 1546300800000 || t > 1830297600000){ continue; }
    for(const m of [2, -2, 1]){
      for(const d of [-3600000, -1, 0, 1800000, 3600000]){
        push(t + d, m);                    // next to the change, shifted away
        const back = addMonths(t + d, -m); // shifted ONTO (the neighbourhood of) the change
        if(!Number.isNaN(back)){ push(back, m); push(back + 1800000, m); push(back - 1800000, m); }
      }
    }
  }
  // month-end overflow, leap years, year wrap
  for(const iso of ['2024-12-31T23:59:59.999Z', '2025-12-31T00:00:00.000Z', '2024-02-29T12:00:00.000Z', '2023-12-30T06:00:00.000Z',
    '2025-01-31T01:02:03.004Z', '2025-03-31T00:00:00.000Z', '2025-08-31T12:00:00.000Z', '1971-01-01T00:00:00.000Z', '2000-02-29T00:00:00.000Z',
    '2011-12-29T12:00:00.000Z', '2011-12-30T12:00:00.000Z', '2011-10-30T12:00:00.000Z']){
    for(const m of [2, 1, 0, -1, -2, 12, 14, -14, 25]){ push(Date.parse(iso), m); }
  }
  // the synthetic corpus range (120 days before T0) and 1972 .. 2097 at random
  for(let i = 0; i < 400; i++){ push(1700000000000 - Math.floor(rnd() * 10368000000), 2); }
  for(let i = 0; i < 400; i++){ push(63072000000 + Math.floor(rnd() * 3.94e12), Math.floor(rnd() * 24) - 12); }
  // keep what lies inside the table's span with a year to spare on either side (outside it the table's first / last offset applies)
  const lo = 31536000000, hi = 4102444800000 - 31536000000;
  const kept = cases.filter(c => c[0] > lo && c[0] < hi && (c[2] === null || (c[2] > lo && c[2] < hi)));
  process.stdout.write(JSON.stringify({transitions: Array.from(table.transitions, Number), offsets: Array.from(table.offsets, Number), cases: kept}));
}else{
  const zones = {};
  for(const tz of ZONES){
    zones[tz] = JSON.parse(execFileSync(process.execPath, [__filename, '--zone'], {env: Object.assign({}, process.env, {TZ: tz}), maxBuffer: 1 << 28}).toString());
  }
  console.log(JSON.stringify({provenance: 'JS engine Date arithmetic (Node ' + process.version + '), one process per TZ: d=new Date(ts); d.setMonth(d.getMonth()+months); ' +
    'd.getTime() — the two calls of sqlProvider.js:999-1009; cases are [ts, months, result|null]; transitions / offsets = the table of ' +
    'sph-pie_amd/host/tzTable.js built in the same process (offsets[0] before transitions[0], offsets[i+1] from transitions[i] on; ms)', zones}));
}
