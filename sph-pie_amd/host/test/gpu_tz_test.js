'use strict';
// GPU check of the retention purge under the PROCESS'S time zone (run with TZ=America/New_York, Australia/Lord_Howe, ...):
// store.purgeRetention(months, now) — month arithmetic done on the device under the transition table host/tzTable.js built from
// this engine's zone rules — must drop exactly the sessions for which the reference's own arithmetic says so:
//   now >= addMonths(createdAt, months),  addMonths = d.setMonth(d.getMonth() + months) on a local Date
// (/root/reference/server/storage/sqlProvider.js:991-1009), evaluated here by the JS engine itself in the same zone.
const assert = require('assert');
const {createStore} = require('../sessionStore');
const {defaultTzTable} = require('../tzTable');

const realNow = Date.now;
let fakeNow = 0;
Date.now = () => fakeNow;
function addMonths(ts, months){ const d = new Date(ts); d.setMonth(d.getMonth() + months); return d.getTime(); }

const table = defaultTzTable();
let s = 0x5EED5EED >>> 0;
const rnd = () => { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; };
const store = createStore();
const made = [];
// sessions created around the zone's transitions of 2023 .. 2025 (two months before each, so that the shifted instant lands on
// or next to the change) and at random over 2022 .. 2025, in order of creation
const stamps = [];
for(const tb of table.transitions){
  const t = Number(tb);
  if(t < 1672531200000 || t > 1767225600000){ continue; }
  for(const d of [-3600000, -1, 0, 1800000, 3600000, 7200000]){ stamps.push(addMonths(t + d, -2)); stamps.push(t + d); }
}
for(let i = 0; i < 3000; i++){ stamps.push(1640995200000 + Math.floor(rnd() * 1.26e11)); }
stamps.sort((a, b) => a - b);
for(const t of stamps){ fakeNow = t; made.push({token: store.createSession('u' + (made.length % 23)).token, createdAt: t}); }
let checks = 0;
for(const now of [1700000000000, 1710054000000 + 3600000, 1730613600000, 1762063200000]){
  fakeNow = now;                                   // (the store itself reads Date.now only for expiry; far future keeps all alive)
  const wantGone = made.filter(m => !m.gone && now >= addMonths(m.createdAt, 2));
  const k = store.purgeRetention(2, now);
  assert.strictEqual(k, wantGone.length, 'purged at ' + now);
  for(const m of wantGone){ m.gone = true; }
  checks++;
}
assert.strictEqual(store.purgeRetention(2, 1762063200000), 0);   // purged rows are tombstones
store.close();
Date.now = realNow;
console.log('host gpu_tz_test ok (' + (process.env.TZ || 'local') + '): ' + made.length + ' sessions, ' + made.filter(m => m.gone).length + ' purged, ' +
  table.transitions.length + ' transitions, ' + checks + ' checks');
