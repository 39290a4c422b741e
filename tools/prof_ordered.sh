#!/bin/bash
# per-kernel times of the ordered-run chain on the dense and the skewed workload -> gpurun_out/prof_ord_*/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=.
for w in "wide:--query wide" "zipf:--users-dist zipf"; do
  tag=${w%%:*}; args=${w#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ord_$tag -o p -- python3 $R/bench.py --steps 40 --warmup 10 --repeat 2 --no-cpu-baseline --no-extra --queries-per-launch 1 $args > $R/gpurun_out/prof_ord_$tag.json 2> $R/gpurun_out/prof_ord_$tag.err || exit 1
done
python3 - <<'PY'
import csv, glob
for tag in ("wide", "zipf"):
    f = glob.glob("gpurun_out/prof_ord_%s/**/*kernel_stats.csv" % tag, recursive=True)
    print("==", tag, f)
    for r in csv.DictReader(open(f[0])):
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "%10.1f us" % (float(r["AverageNs"]) / 1e3))
PY
