#!/usr/bin/env node
// TEST INFRASTRUCTURE — golden-vector generator. Runs ONLY in the build container, never on the GPU box.
//
// Drives the real reference module /root/reference/server/sessionStore.js (the one hot-path file that
// is importable on Node 12, SURVEY.md §8c) with Date.now stubbed, and records inputs + observed outputs
// as small JSON fixtures under tests/golden/.  Only data (timestamps, user ids, survivor sets) is
// written; no reference source text is copied.
//
//   G1 liveness   getSession()            sessionStore.js:21-35   pins  live  <=>  expiresAt >  now
//   G2 purge      purgeExpiredSessions()  sessionStore.js:66-73   pins  scan with a single `now`
//   G3 user match deleteSessionsForUser() sessionStore.js:55-64   pins  userId === x, falsy no-op
//   G4 touch      touchSession()          sessionStore.js:37-45   pins  end' = now + TTL, start kept
//
// usage: node oracle/gen_golden.js [/root/reference] [tests/golden]
'use strict';
const fs = require('fs');
const path = require('path');

const refRoot = process.argv[2] || '/root/reference';
const outDir = process.argv[3] || path.join(__dirname, '..', 'tests', 'golden');
const modPath = path.join(refRoot, 'server', 'sessionStore.js');

const realNow = Date.now;
let fakeNow = 0;
Date.now = function(){ return fakeNow; };

function fresh(){
  delete require.cache[require.resolve(modPath)];
  return require(modPath);
}

// deterministic corpus: 72 sessions over 5 users, creation times spread over ~30 h so that a sweep of
// `now` crosses many expiry edges. splitmix-free: a small LCG is enough for a fixture.
const T0 = 1700000000000;
const USERS = ['u-alpha', 'u-bravo', 'u-charlie', 'u-delta', 'u-echo'];
function lcg(seed){ let s = seed >>> 0; return function(){ s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s; }; }

function buildCorpus(store){
  const rnd = lcg(0x5EED5EED);
  const rows = [];
  let t = T0 - 30 * 3600 * 1000;
  for(let i = 0; i < 72; i++){
    t += 1 + (rnd() % (50 * 60 * 1000));          // strictly increasing createdAt, like a live server
    if(i % 9 === 4){ t -= 0; }                      // keep a few equal-gap neighbours
    fakeNow = t;
    const user = USERS[rnd() % USERS.length];
    const made = store.createSession(user);
    rows.push({row: i, user: user, createdAt: t, expiresAt: made.expiresAt, token: made.token});
  }
  // two sessions created in the same millisecond (ties on start) for two different users and one same user
  fakeNow = t + 1000;
  for(const user of [USERS[0], USERS[1], USERS[0]]){
    const made = store.createSession(user);
    rows.push({row: rows.length, user: user, createdAt: fakeNow, expiresAt: made.expiresAt, token: made.token});
  }
  return rows;
}

function publicRows(rows){
  return rows.map(r => ({row: r.row, user: r.user, createdAt: r.createdAt, expiresAt: r.expiresAt}));
}

const out = {
  provenance: 'outputs of /root/reference/server/sessionStore.js driven by oracle/gen_golden.js with Date.now stubbed (Node ' + process.version + ')',
  ttl_ms: null, cookie_name: null, users: USERS, sessions: null, G1: [], G2: [], G3: [], G4: []
};

// ---- G1: liveness sweep (fresh store per `now`, because getSession deletes what it finds dead) ----
{
  let store = fresh();
  out.ttl_ms = store.SESSION_TTL_MS;
  out.cookie_name = store.SESSION_COOKIE_NAME;
  let rows = buildCorpus(store);
  out.sessions = publicRows(rows);
  const nows = new Set();
  for(const k of [3, 17, 40, 71, 73]){
    const e = rows[k].expiresAt;
    nows.add(e - 1); nows.add(e); nows.add(e + 1);
  }
  nows.add(rows[0].createdAt); nows.add(rows[rows.length - 1].expiresAt + 5); nows.add(T0 - 6 * 3600 * 1000);
  for(const now of Array.from(nows).sort((a, b) => a - b)){
    store = fresh(); rows = buildCorpus(store);
    fakeNow = now;
    const live = rows.map(r => {
      const s = store.getSession(r.token);
      if(s !== null){
        if(s.userId !== r.user || s.createdAt !== r.createdAt || s.expiresAt !== r.expiresAt){ throw new Error('schema drift'); }
      }
      return s !== null ? 1 : 0;
    });
    out.G1.push({now: now, live: live});
  }
  // falsy token -> null (sessionStore.js:22-24)
  out.G1_falsy_token_null = [store.getSession(''), store.getSession(null), store.getSession(undefined)].every(v => v === null);
}

// ---- G2: purge at chosen `now` values; survivors observed at the SAME stubbed now ----
for(const pick of [10, 40, 60]){
  const store = fresh(); const rows = buildCorpus(store);
  const now = rows[pick].expiresAt;               // exactly on an edge: that row must be purged (<=)
  fakeNow = now;
  store.purgeExpiredSessions();
  const survivors = rows.filter(r => store.getSession(r.token) !== null).map(r => r.row);
  out.G2.push({now: now, survivors: survivors});
}

// ---- G3: deleteSessionsForUser, incl. falsy userId no-op; observe with now before any expiry ----
for(const user of [USERS[0], USERS[3], 'nobody', '', null]){
  const store = fresh(); const rows = buildCorpus(store);
  fakeNow = rows[0].createdAt;                     // nothing has expired yet at this instant
  store.deleteSessionsForUser(user);
  const survivors = rows.filter(r => store.getSession(r.token) !== null).map(r => r.row);
  out.G3.push({user: user, observe_now: fakeNow, survivors: survivors});
}

// ---- G4: touch re-arms end = now + TTL and keeps start; dead sessions are not revived ----
{
  const store = fresh(); const rows = buildCorpus(store);
  const touches = [];
  for(const k of [5, 30, 70]){
    const now = rows[k].expiresAt - 1;              // still live by 1 ms
    fakeNow = now;
    const res = store.touchSession(rows[k].token);
    const after = store.getSession(rows[k].token);
    touches.push({row: k, now: now, returned: res, after: {userId: after.userId, createdAt: after.createdAt, expiresAt: after.expiresAt}});
  }
  const k = 2; fakeNow = rows[k].expiresAt;        // dead exactly at the edge
  const res = store.touchSession(rows[k].token);
  touches.push({row: k, now: fakeNow, returned: res, after: null});
  out.G4 = touches;
}

Date.now = realNow;
fs.mkdirSync(outDir, {recursive: true});
const file = path.join(outDir, 'sessionstore_g1_g4.json');
fs.writeFileSync(file, JSON.stringify(out, null, 1) + '\n');
console.log('wrote', file, 'sessions=' + out.sessions.length, 'G1=' + out.G1.length, 'G2=' + out.G2.length, 'G3=' + out.G3.length, 'G4=' + out.G4.length);
