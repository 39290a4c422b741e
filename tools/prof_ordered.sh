#!/bin/bash
# per-kernel times of the ordered-run chains (dense query, Zipf single, Zipf batch) -> gpurun_out/prof_ord_*/ ; usage: tools/prof_ordered.sh [tags]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in "wide:--query wide --queries-per-launch 1" "zipf:--users-dist zipf --queries-per-launch 1" "zipfq16:--users-dist zipf"; do
  tag=${w%%:*}; args=${w#*:}
  case " ${*:-wide zipf zipfq16} " in *" $tag "*) ;; *) continue;; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ord_$tag -o p -- python3 bench.py --steps 40 --warmup 12 --repeat 2 --no-cpu-baseline --no-extra $args > gpurun_out/prof_ord_$tag.json 2> gpurun_out/prof_ord_$tag.err || exit 1
done
python3 - <<'PY'
import csv, glob
for tag in ("wide", "zipf", "zipfq16"):
    f = glob.glob("gpurun_out/prof_ord_%s/**/*kernel_stats.csv" % tag, recursive=True)
    if not f: continue
    print("==", tag)
    for r in csv.DictReader(open(f[0])):
        if int(r["Calls"]) >= 20: print(r["Name"][:80].ljust(80), r["Calls"].rjust(5), "%10.1f us" % (float(r["AverageNs"]) / 1e3))
PY
