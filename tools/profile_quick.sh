#!/bin/bash
# kernel-trace statistics of the metric step (bench.py, 100 steps) -> average duration per kernel.  Usage on the GPU box: bash tools/profile_quick.sh <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/quick_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats -f csv -d $OUT -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs > $OUT/bench.json 2> $OUT/bench.err || echo "failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print("%-50s %5s %9.2f us" % (r["Name"].replace("(anonymous namespace)::", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
