#!/bin/bash
# keyed table pass: unroll x grid sweep through bench.py (one process per point); results -> gpurun_out/keyed_sweep.txt
out=gpurun_out/keyed_sweep.txt
mkdir -p gpurun_out
: > $out
run() {
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel'], r['kernel_variant'], r['kernel_ms'], r['k1_blocks'])" >> $out
}
for v in ${KEYED_FORMS:-0x485 0x405}; do
  for b in ${KEYED_BLOCKS:-256 512 768 1024 1536 2048 4096 8192}; do
    echo "== keyed=$v blocks=$b" >> $out
    PIE_K1_KEYED=$v PIE_K1_BLOCKS_KEYED=$b run || exit 1
  done
done
echo "== keyed off" >> $out
PIE_K1_KEYED=0 run
