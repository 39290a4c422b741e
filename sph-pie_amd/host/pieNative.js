'use strict';
// Loads the raw N-API addon (csrc/pie_napi.c) and opens libpie_hip.so through it.  There is no JS or CPU
// fallback for the scan: if either piece is missing this throws, loudly, at require time of the caller.
const path = require('path');
const fs = require('fs');

const ADDON = path.join(__dirname, 'pie_napi.node');
const LIB = path.join(__dirname, '..', 'libpie_hip.so');

let native = null;

function load(){
  if(native){
    return native;
  }
  if(!fs.existsSync(ADDON)){
    throw new Error('pie_napi.node is not built (run: python -c "import __graft_entry__ as g; g.build()"); no CPU fallback exists');
  }
  const addon = require(ADDON);
  const abi = addon.open(LIB);
  if(abi !== 1){
    throw new Error('libpie_hip.so ABI ' + abi + ' does not match this host (expected 1)');
  }
  native = addon;
  return native;
}

module.exports = {load, ADDON, LIB};
