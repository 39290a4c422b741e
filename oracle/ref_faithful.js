#!/usr/bin/env node
// TEST INFRASTRUCTURE — reference-faithful JavaScript restatement (baseline "B1" of BASELINE.md §3).
// Keeps the reference's own data structure and loops: a Map of {userId, createdAt, expiresAt} objects keyed by
// a token hash, iterated in insertion order like /root/reference/server/sessionStore.js:59-63,68-72, then per
// user the window filter (server/storage/sqlProvider.js:284) and ascending-start order (:276, stable sort so
// ties keep insertion order).  Single-threaded, as the reference is.  Used (1) by the Node GPU tests as the
// checker for the host modules and (2) by bench.py as the JS CPU baseline (--bench).
'use strict';

const MASK64 = (1n << 64n) - 1n;
const GOLDEN = 0x9E3779B97F4A7C15n;
function mix64(z){
  z = ((z ^ (z >> 30n)) * 0xBF58476D1CE4E5B9n) & MASK64;
  z = ((z ^ (z >> 27n)) * 0x94D049BB133111EBn) & MASK64;
  return z ^ (z >> 31n);
}
const smOut = (seed, n) => mix64((seed + (n + 1n) * GOLDEN) & MASK64);
const mulhi = (a, b) => (a * b) >> 64n;

const T0 = 1700000000000, SPAN = 10368000000n, TTL = 43200000;

// same corpus as oracle/pie_oracle.c:pie_oracle_gen (auth variant, random order); BigInt arithmetic is slow,
// so this is for small N; the --bench mode uses a cheap float PRNG instead (timing only).
function genCorpus(seed, n, U, D){
  const rows = [];
  for(let i = 0n; i < BigInt(n); i++){
    const r0 = smOut(seed, 4n * i), r1 = smOut(seed, 4n * i + 1n), r2 = smOut(seed, 4n * i + 2n);
    const start = T0 - Number(mulhi(r2, SPAN));
    rows.push({user: Number(mulhi(r0, BigInt(U))), disc: Number(mulhi(r1, BigInt(D))), start, end: start + TTL});
  }
  return rows;
}

// sessions: Map in insertion order; value {userId, createdAt, expiresAt, disc}
function buildMap(rows){
  const sessions = new Map();
  rows.forEach((r, i) => sessions.set('t' + i, {userId: r.user, createdAt: r.start, expiresAt: r.end, disc: r.disc, row: i}));
  return sessions;
}

// the [DERIVED] scan of SURVEY.md §8 a-D, written the way the reference writes its loops
function scanFeeds(sessions, nUsers, now, cutoff, discAllowed){
  const feeds = [];
  for(let u = 0; u < nUsers; u++){ feeds.push([]); }
  for(const [, session] of sessions.entries()){
    if(session.expiresAt <= now){ continue; }                                  // sessionStore.js:30,69
    if(!(Number.isFinite(session.createdAt) && session.createdAt >= cutoff)){ continue; }  // sqlProvider.js:284
    if(!discAllowed(session.disc)){ continue; }                                // disciplineConfig.js:88-97
    feeds[session.userId].push(session);                                       // sessionStore.js:60 (userId === x)
  }
  for(const feed of feeds){
    feed.sort((a, b) => a.createdAt - b.createdAt);                            // ORDER BY start_ts ASC; stable
  }
  return feeds;
}

function bench(n, U, D){
  let s = 0x5EED5EED >>> 0;
  const rnd = () => { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; };
  const sessions = new Map();
  for(let i = 0; i < n; i++){
    const start = T0 - Math.floor(rnd() * 10368000000);
    sessions.set('t' + i, {userId: Math.floor(rnd() * U), createdAt: start, expiresAt: start + TTL, disc: Math.floor(rnd() * D), row: i});
  }
  const now = T0 - 6 * 3600 * 1000, cutoff = T0 - 61 * 86400 * 1000;
  const allowed = d => d >= 0 && d < D && (d % 2 === 0);
  let reps = 0, selected = 0;
  const t0 = process.hrtime.bigint();
  let dt = 0;
  do{
    const feeds = scanFeeds(sessions, U, now, cutoff, allowed);
    selected = feeds.reduce((acc, f) => acc + f.length, 0);
    reps++;
    dt = Number(process.hrtime.bigint() - t0) / 1e9;
  }while(dt < 3 && reps < 100);
  console.log(JSON.stringify({kind: 'reference-faithful JS (Map of session objects), 1 thread', node: process.version, rows: n, users: U,
    reps, seconds: dt, sessions_per_sec: n * reps / dt, selected}));
}

if(require.main === module){
  const a = process.argv.slice(2);
  if(a[0] === '--bench'){
    bench(Number(a[1] || 1000000), Number(a[2] || 1000), Number(a[3] || 32));
  }
}

module.exports = {genCorpus, buildMap, scanFeeds, mix64, smOut};
