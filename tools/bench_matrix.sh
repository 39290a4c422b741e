#!/bin/bash
# secondary workloads of DESIGN.md section 6, same binary; -> gpurun_out/bench_matrix.txt
out=gpurun_out/bench_matrix.txt
mkdir -p gpurun_out
: > $out
run() {
  echo "== $*" >> $out
  python3 bench.py --steps 100 --warmup 10 --repeat 3 --no-cpu-baseline --no-extra "$@" 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms_per_step', round(d['ms_per_step'],5), 'Q', d.get('queries_per_launch'), 'feeds/s %.3e' % d['value'], r['kernel'], r['kernel_variant'], 'k1_ms', round(r['kernel_ms'],5), 'M', d['config']['selected_rows_rank0'], d.get('mixed') and {k: round(v, 4) for k, v in d['mixed'].items() if k.endswith('_ms')} or '')" >> $out
}
run
run --queries-per-launch 16
run --queries-per-launch 1
run --order time
run --order time --queries-per-launch 16
run --order time --queries-per-launch 1
run --order clustered --queries-per-launch 1
run --variant interval --queries-per-launch 1
run --users-dist zipf --queries-per-launch 1
run --users-dist zipf --queries-per-launch 16
run --users-dist zipf
run --rows 10000000 --users 10000
run --rows 10000000 --users 10000 --queries-per-launch 16
run --rows 10000000 --users 10000 --queries-per-launch 1
run --rows 12500000 --users 12500
run --rows 12500000 --users 12500 --queries-per-launch 16
run --rows 1000 --users 10 --disc 4 --queries-per-launch 1
run --query wide --steps 30 --queries-per-launch 1
run --mode expired --steps 60
run --mode mixed --steps 50
echo "== --mode archive" >> $out
python3 bench.py --mode archive --steps 10 --warmup 2 --repeat 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms_per_step', round(d['ms_per_step'],4), 'chain_ms', round(r['chain_ms'],4), 'frac', round(r['frac'],3), 'queued', d['config']['queued_rows'])" >> $out
cat $out
