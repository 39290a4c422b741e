'use strict';
// The time-zone transition table the device needs for calendar-month arithmetic on LOCAL dates (pie_retention_purge_tz:
// /root/reference/server/storage/sqlProvider.js:998-1009 is `date.setMonth(date.getMonth() + months)` on a local-time Date,
// and local time is not a fixed offset from UTC where daylight saving applies).  Built from the JS engine's own zone rules —
// the very rules the reference's Date calls use in this process — by probing local fields day by day and bisecting every
// change to the millisecond.
//   offsets[0] applies before transitions[0]; offsets[i + 1] from transitions[i] on (milliseconds, local = UTC + offset).

function offsetAt(t){
  const d = new Date(t);
  return Date.UTC(d.getFullYear(), d.getMonth(), d.getDate(), d.getHours(), d.getMinutes(), d.getSeconds(), d.getMilliseconds()) - t;
}

const DAY = 86400000;
const cache = new Map();

function buildTzTable(fromMs, toMs){
  const key = (process.env.TZ || '') + ':' + fromMs + ':' + toMs;
  if(cache.has(key)){ return cache.get(key); }
  const transitions = [];
  const offsets = [offsetAt(fromMs)];
  let prevT = fromMs, prevOff = offsets[0];
  for(let t = fromMs + DAY; prevT < toMs; t += DAY){
    const at = Math.min(t, toMs);
    const off = offsetAt(at);
    if(off !== prevOff){
      let lo = prevT, hi = at;                       // offset(lo) === prevOff, offset(hi) === off
      while(hi - lo > 1){
        const mid = lo + Math.floor((hi - lo) / 2);
        if(offsetAt(mid) === prevOff){ lo = mid; }else{ hi = mid; }
      }
      transitions.push(hi);
      offsets.push(off);
      prevOff = off;
    }
    prevT = at;
  }
  const table = {transitions: BigInt64Array.from(transitions.map(BigInt)), offsets: BigInt64Array.from(offsets.map(BigInt))};
  cache.set(key, table);
  return table;
}

// 1970-01-01 .. 2100-01-01: the span the golden vectors cover; outside it the first / last offset applies
function defaultTzTable(){ return buildTzTable(0, 4102444800000); }

module.exports = {buildTzTable, defaultTzTable, offsetAt};
