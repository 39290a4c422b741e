#!/bin/bash
# keyed table pass: form x grid sweep through bench.py (one process per point); results -> gpurun_out/keyed_sweep.txt
# KEYED_FORMS: values for PIE_K1_KEYED (0x485 2-byte key, 0xC85 1-byte top-of-range key); KEYED_BLOCKS: grid sizes
out=gpurun_out/keyed_sweep.txt
mkdir -p gpurun_out
: > $out
run() {
  python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel'], r['kernel_variant'], r['kernel_ms'], r['k1_blocks'])" >> $out
}
for v in ${KEYED_FORMS:-0xC85 0xC05}; do
  for b in ${KEYED_BLOCKS:-512 1024 2048 4096}; do
    echo "== keyed=$v blocks=$b" >> $out
    PIE_K1_KEYED=$v PIE_K1_BLOCKS_KEYED=$b PIE_K1_BLOCKS_FINE=$b run || exit 1
  done
done
echo "== 2-byte key" >> $out
PIE_K1_KEYED=0x485 run
