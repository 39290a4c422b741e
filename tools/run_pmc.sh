#!/bin/bash
# Run on the GPU box (via gpurun): PMC passes for the scan kernels, each in its own rocprofv3 run (counters only with
# --kernel-trace, as the pool requires), then a kernel-trace --stats run of the same command.  The command is bench.py's
# default workload, so one set of passes covers the headline kernel AND the every-byte form of roofline_full_read.
# The stats run pins ONE batch lane (PIE_BATCH_LANES=1): a kernel's trace duration is then the kernel alone on the chip — the
# figure bench.py's roofline.kernel_ms reports — not its duration while it shares the chip with the other lanes' launches; the
# run with the default lanes is kept beside it (<tag>_stats_lanes).
# usage: [TRAFFIC_OUT=traffic_x.json TRAFFIC_WORKLOAD="..."] tools/run_pmc.sh <tag> [extra bench.py flags]    -> gpurun_out/<tag>_{fetch,write,rdreq,stats}/... + profiles/traffic.json
set -o pipefail
tag=${1:-pmc}
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cmd="python3 bench.py --steps 6 --warmup 3 --repeat 1 --no-cpu-baseline --no-mixed-leg $*"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_fetch -- $cmd > gpurun_out/${tag}_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_write -- $cmd > gpurun_out/${tag}_write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d gpurun_out/${tag}_rdreq -- $cmd > gpurun_out/${tag}_rdreq.log 2>&1 &&
PIE_BATCH_LANES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --steps 30 --warmup 5 --repeat 2 --no-cpu-baseline --no-mixed-leg $* > gpurun_out/${tag}_stats.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_lanes -- python3 bench.py --steps 30 --warmup 5 --repeat 2 --no-cpu-baseline --no-mixed-leg $* > gpurun_out/${tag}_stats_lanes.log 2>&1 &&
python3 tools/pmc_traffic.py gpurun_out ${tag} ${TRAFFIC_OUT:-traffic.json} "${TRAFFIC_WORKLOAD:-bench.py default: 1e8 sessions / 1e5 users / 32 disciplines, random order, auth variant, spec query}"
