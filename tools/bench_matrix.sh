#!/bin/bash
# secondary workloads of DESIGN.md section 6, same binary; -> gpurun_out/bench_matrix.txt
out=gpurun_out/bench_matrix.txt
mkdir -p gpurun_out
: > $out
run() {
  echo "== $*" >> $out
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms_per_step', round(d['ms_per_step'],5), r['kernel'], r['kernel_variant'], 'k1_ms', round(r['kernel_ms'],5), 'M', d['config']['selected_rows_rank0'])" >> $out
}
run
run --order clustered
run --variant interval
run --users-dist zipf
run --rows 10000000 --users 10000
run --rows 1000 --users 10 --disc 4
run --query wide --steps 50
run --mode expired --steps 100
