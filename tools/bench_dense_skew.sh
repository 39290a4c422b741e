#!/bin/bash
# the dense ("wide") and skewed (Zipf) workloads with and without the ordered run, same binary; -> gpurun_out/bench_dense_skew.txt
out=gpurun_out/bench_dense_skew.txt
mkdir -p gpurun_out
: > $out
run() {
  echo "== PIE_ORDERED=${PIE_ORDERED:-1} $*" >> $out
  python3 bench.py --steps 60 --warmup 10 --repeat 3 --no-cpu-baseline --no-extra "$@" 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms_per_step', round(d['ms_per_step'],5), 'Q', d.get('queries_per_launch'), 'feeds/s %.3e' % d['value'], r['kernel'], r['kernel_variant'], 'k1_ms', round(r['kernel_ms'],5), 'M', d['config']['selected_rows_rank0'], 'ordered', d['index']['ordered_run'])" >> $out
}
run --query wide --queries-per-launch 1
PIE_ORDERED=0 run --query wide --queries-per-launch 1
run --users-dist zipf --queries-per-launch 1
PIE_ORDERED=0 run --users-dist zipf --queries-per-launch 1
run --users-dist zipf
PIE_ORDERED=2 run --queries-per-launch 1
run --queries-per-launch 1
PIE_ORDERED=2 run --query wide --order clustered --queries-per-launch 1
PIE_ORDERED=2 run --order time --queries-per-launch 1
cat $out
