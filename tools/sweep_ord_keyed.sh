#!/bin/bash
# grid x unroll of the ordered run's key-stream kernels on the Zipf corpus (single query and 16-query batch); the unroll
# dimension needs the PIE_ORD_UNROLL dispatch of commit 88fbee3 (removed afterwards: it made no difference)
for u in 2 4 8; do for g in 4 6 8 10 12; do
  for q in "--queries-per-launch 1" ""; do
    PIE_ORD_UNROLL=$u PIE_ORD_GRID=$g python3 bench.py --steps 40 --warmup 12 --repeat 2 --no-cpu-baseline --no-extra --users-dist zipf $q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('unroll $u grid $g Q', d.get('queries_per_launch'), 'ms_per_step %.5f' % d['ms_per_step'], 'k1_ms %.5f' % r['kernel_ms'])"
  done
done; done
