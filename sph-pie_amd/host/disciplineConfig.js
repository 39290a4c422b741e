'use strict';
// Host mirror of the reference's discipline table module (same export names and results as
// /root/reference/server/disciplineConfig.js:148-160) plus the two helpers the device path needs:
// disciplineIndex() (id -> int32 column value) and disciplineMask() (ids -> 64-bit predicate table).
// Own implementation: a lookup Map built once at load, instead of repeated linear finds.
const fs = require('fs');
const path = require('path');

const TABLE_FILE = process.env.PIE_DISCIPLINES_JSON || path.join(__dirname, 'config', 'disciplines.json');

const lower = v => (typeof v === 'string' ? v.trim().toLowerCase() : '');

function readTable(file){
  // throws when the file is missing or invalid, like the reference does at require time (:7-8)
  const doc = JSON.parse(fs.readFileSync(file, 'utf8'));
  const levels = (Array.isArray(doc.roles) ? doc.roles : []).map(lower).filter(Boolean);
  const rows = [];
  for(const raw of (Array.isArray(doc.disciplines) ? doc.disciplines : [])){
    if(raw === null || typeof raw !== 'object'){
      continue;
    }
    const id = lower(raw.id);
    const name = typeof raw.name === 'string' ? raw.name.trim() : '';
    if(id && name){                                   // entries lacking id or name are dropped (:24-26)
      rows.push({id, name, default: Boolean(raw.default), forms: Boolean(raw.forms)});
    }
  }
  return {levels, rows};
}

const loaded = readTable(TABLE_FILE);
const ROLE_LEVELS = loaded.levels;
const DISCIPLINES = loaded.rows;
const byId = new Map();
DISCIPLINES.forEach((d, i) => { if(!byId.has(d.id)){ byId.set(d.id, i); } });   // first match wins, as Array.find does
const DEFAULT_DISCIPLINE = DISCIPLINES.find(d => d.default) || DISCIPLINES[0] || null;

function disciplineIndex(id){
  const key = lower(id);
  return key && byId.has(key) ? byId.get(key) : -1;
}

function findDiscipline(id){
  const i = disciplineIndex(id);
  return i < 0 ? null : DISCIPLINES[i];
}

function getRoleKey(disciplineId, level){
  const d = findDiscipline(disciplineId);
  const lv = lower(level);
  return d && ROLE_LEVELS.indexOf(lv) >= 0 ? d.id + '.' + lv : null;
}

function listRoleKeys(){
  const out = [];
  DISCIPLINES.forEach(d => ROLE_LEVELS.forEach(lv => out.push(d.id + '.' + lv)));
  return out;
}

const ALIAS_LEVEL = {lead: 'lead', operator: 'operator', stagecrew: 'crew', crew: 'crew'};

function normalizeRole(role){
  if(typeof role !== 'string'){
    return null;
  }
  const text = role.trim();
  if(text === ''){
    return null;
  }
  const key = text.toLowerCase();
  if(key === 'admin'){
    return 'admin';
  }
  if(Object.prototype.hasOwnProperty.call(ALIAS_LEVEL, key)){
    return getRoleKey(DEFAULT_DISCIPLINE ? DEFAULT_DISCIPLINE.id : undefined, ALIAS_LEVEL[key]) || null;
  }
  if(text.indexOf('.') < 0){
    return null;
  }
  const parts = text.split('.');                      // only the first two parts count, like the destructuring at :83
  return getRoleKey(parts[0], parts[1]);
}

function parseRoleKey(roleKey){
  const key = lower(roleKey);
  if(typeof roleKey !== 'string' || key === ''){
    return null;
  }
  if(key === 'admin'){
    return {disciplineId: null, level: 'admin'};
  }
  const parts = key.split('.');
  if(parts.length !== 2 || ROLE_LEVELS.indexOf(parts[1]) < 0){
    return null;
  }
  const d = findDiscipline(parts[0]);
  return d ? {disciplineId: d.id, level: parts[1]} : null;
}

function roleMatchesLevel(roleKey, level){
  const p = parseRoleKey(roleKey);
  return Boolean(p && p.level === level);
}

function roleMatchesDiscipline(roleKey, disciplineId){
  const p = parseRoleKey(roleKey);
  return Boolean(p && p.disciplineId === disciplineId);
}

function getDisplayName(roleKey){
  if(roleKey === 'admin'){
    return 'Admin';
  }
  const p = parseRoleKey(roleKey);
  if(!p){
    return roleKey;
  }
  const d = findDiscipline(p.disciplineId);
  const levelName = p.level.charAt(0).toUpperCase() + p.level.slice(1);
  return (d ? d.name : p.disciplineId) + ' ' + levelName;
}

// ids (or '*' / undefined for every discipline) -> BigInt bit mask over DISCIPLINES indices; unknown ids add nothing
function disciplineMask(ids){
  let mask = 0n;
  if(ids === undefined || ids === null || ids === '*'){
    for(let i = 0; i < DISCIPLINES.length; i++){ mask |= 1n << BigInt(i); }
    return mask;
  }
  for(const id of (Array.isArray(ids) ? ids : [ids])){
    const i = disciplineIndex(id);
    if(i >= 0){ mask |= 1n << BigInt(i); }
  }
  return mask;
}

module.exports = {
  ROLE_LEVELS, DISCIPLINES, DEFAULT_DISCIPLINE,
  getRoleKey, listRoleKeys, normalizeRole, parseRoleKey, roleMatchesLevel, roleMatchesDiscipline, getDisplayName, findDiscipline,
  disciplineIndex, disciplineMask
};
