#!/usr/bin/env node
// TEST INFRASTRUCTURE — golden-trace generator. Runs ONLY in the build container, never on the GPU box.
//
// G5: a long, seeded, randomised sequence of calls into the real /root/reference/server/sessionStore.js (Date.now
// stubbed), recording every call and what the reference answered.  Where G1-G4 pin one predicate each on a fixed
// corpus, G5 pins their interplay: sessions created in the same millisecond, touched, deleted one by one and per
// user, purged, and looked up after they died (getSession drops a dead session as a side effect, sessionStore.js:30-33).
// Only data is written (timestamps, user names, token ORDINALS — tokens themselves are random in the reference).
//
// usage: node oracle/gen_trace_golden.js [/root/reference] [tests/golden]
'use strict';
const fs = require('fs');
const path = require('path');

const refRoot = process.argv[2] || '/root/reference';
const outDir = process.argv[3] || path.join(__dirname, '..', 'tests', 'golden');
const store = require(path.join(refRoot, 'server', 'sessionStore.js'));

const realNow = Date.now;
let now = 1700000000000;
Date.now = () => now;
let s = 0xC0FFEE >>> 0;
const rnd = () => { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s; };
const pick = n => rnd() % n;

const USERS = [];
for(let i = 0; i < 12; i++){ USERS.push('user-' + String.fromCharCode(97 + i)); }
const tokens = [];          // ordinal -> token string (never written out)
const ops = [];
const pub = r => (r === null ? null : {userId: r.userId, createdAt: r.createdAt, expiresAt: r.expiresAt});

function census(){
  const live = [];
  tokens.forEach((t, i) => { if(store.getSession(t) !== null){ live.push(i); } });
  ops.push({op: 'census', now, live});
}

for(let step = 0; step < 1600; step++){
  const p = pick(100);
  if(p < 42){
    const user = USERS[pick(USERS.length)];
    const made = store.createSession(user);
    tokens.push(made.token);
    ops.push({op: 'create', now, user, expiresAt: made.expiresAt});
  } else if(p < 60 && tokens.length){
    const tok = pick(tokens.length);
    ops.push({op: 'get', now, tok, result: pub(store.getSession(tokens[tok]))});
  } else if(p < 70 && tokens.length){
    const tok = pick(tokens.length);
    const r = store.touchSession(tokens[tok]);
    ops.push({op: 'touch', now, tok, result: r === null ? null : {userId: r.userId, expiresAt: r.expiresAt}});
  } else if(p < 74 && tokens.length){
    const tok = pick(tokens.length);
    store.deleteSession(tokens[tok]);
    ops.push({op: 'del', now, tok});
  } else if(p < 77){
    const q = pick(10);
    const user = q === 0 ? '' : (q === 1 ? 'nobody' : USERS[pick(USERS.length)]);
    store.deleteSessionsForUser(user);
    ops.push({op: 'delUser', now, user});
  } else if(p < 82){
    store.purgeExpiredSessions();
    ops.push({op: 'purge', now});
  } else if(p < 85){
    census();
  } else {
    // time moves: often not at all (same-millisecond neighbours), usually minutes, sometimes hours
    const q = pick(10);
    now += q < 2 ? 0 : (q < 8 ? 1 + pick(40 * 60 * 1000) : pick(7 * 3600 * 1000));
  }
}
now += 1; census();
now += 13 * 3600 * 1000; census();     // everything untouched for 13 h is dead

Date.now = realNow;
const out = {
  provenance: 'calls into /root/reference/server/sessionStore.js and its answers, recorded by oracle/gen_trace_golden.js with Date.now stubbed (Node ' + process.version + ')',
  ttl_ms: store.SESSION_TTL_MS, users: USERS, sessions: tokens.length, ops
};
fs.mkdirSync(outDir, {recursive: true});
const file = path.join(outDir, 'sessionstore_g5_trace.json');
fs.writeFileSync(file, JSON.stringify(out) + '\n');
const kinds = {};
ops.forEach(o => { kinds[o.op] = (kinds[o.op] || 0) + 1; });
console.log('wrote', file, 'sessions=' + tokens.length, JSON.stringify(kinds), fs.statSync(file).size + ' bytes');
