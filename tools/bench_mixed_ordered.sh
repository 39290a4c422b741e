#!/bin/bash
# mixed workload (create + touch + scan per step) with the ordered run never / always, uniform and Zipf users -> gpurun_out/bench_mixed_ordered.txt
out=gpurun_out/bench_mixed_ordered.txt
mkdir -p gpurun_out
: > $out
run() {
  echo "== PIE_ORDERED=${PIE_ORDERED:-1} $*" >> $out
  python3 bench.py --mode mixed --steps 60 --warmup 10 --repeat 3 --no-cpu-baseline --no-extra "$@" 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms_per_step', round(d['ms_per_step'],5), r['kernel'], r['kernel_variant'], 'M', d['config']['selected_rows_rank0'], {k: round(v, 4) for k, v in d['mixed'].items() if k.endswith('_ms')}, 'ordered', d['index']['ordered_run'])" >> $out
}
PIE_ORDERED=0 run --mixed-clock end
run --mixed-clock end
PIE_ORDERED=0 run --mixed-clock end --users-dist zipf
run --mixed-clock end --users-dist zipf
PIE_ORDERED=0 run --mixed-clock end --query wide
run --mixed-clock end --query wide
run --query wide
run --users-dist zipf
cat $out
