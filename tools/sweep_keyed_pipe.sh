#!/bin/bash
# single-query keyed table pass: pipelined candidate evaluation (variant bit 0x10) x grid size; K1 kernel time from HIP events
out=gpurun_out/sweep_keyed_pipe.txt
: > $out
for v in 0xC85 0xC95; do for b in 256 512 1024 2048 4096; do
  echo "== PIE_K1_KEYED=$v PIE_K1_BLOCKS_FINE=$b" >> $out
  PIE_K2_RIDE=0 PIE_K1_KEYED=$v PIE_K1_BLOCKS_FINE=$b python3 bench.py --steps 60 --warmup 10 --repeat 3 --no-cpu-baseline --no-extra --queries-per-launch 1 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms_per_step', round(d['ms_per_step'],5), r['kernel'], 'k1_ms', round(r['kernel_ms'],5), 'blocks', r['k1_blocks'])" >> $out
done; done
cat $out
