#!/bin/bash
# A/B: interleave run length of the keyed / batched table pass (PIE_CHUNK_RUN builds under build/ab) x row order x Q
out=gpurun_out/ab_chunk_run.txt
: > $out
for r in 1 2 4 8; do for order in random time; do for q in 1 16; do
  echo "== run=$r order=$order Q=$q" >> $out
  PIE_HIP_LIB=$PWD/build/ab/libpie_r$r.so python3 bench.py --steps 100 --warmup 10 --repeat 3 --no-cpu-baseline --no-extra --order $order --queries-per-launch $q 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms_per_step', round(d['ms_per_step'],5), 'k1_ms', round(r['kernel_ms'],5))" >> $out
done; done
  echo "== run=$r mixed" >> $out
  PIE_HIP_LIB=$PWD/build/ab/libpie_r$r.so python3 bench.py --steps 50 --warmup 5 --repeat 3 --no-cpu-baseline --no-extra --mode mixed 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms_per_step', round(d['ms_per_step'],5), 'k1_ms', round(r['kernel_ms'],5))" >> $out
done
cat $out
