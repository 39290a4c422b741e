#!/bin/bash
# Run on the GPU box (via gpurun): PMC passes for the scan kernel, each in its own rocprofv3 run
# (counters only with --kernel-trace, as the pool requires), then a kernel-trace --stats run of the same command.
# usage: tools/run_pmc.sh <tag>     -> gpurun_out/<tag>_{fetch,write,rdreq,stats}/...
set -o pipefail
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cmd="python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_fetch -- $cmd > gpurun_out/${tag}_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_write -- $cmd > gpurun_out/${tag}_write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d gpurun_out/${tag}_rdreq -- $cmd > gpurun_out/${tag}_rdreq.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_stats.log 2>&1 &&
python3 tools/pmc_traffic.py gpurun_out ${tag}
