#!/bin/bash
# Instruction-fetch side of the fused ODE/ELBO kernel (51 KB of straight-line code per workgroup): I-cache requests / misses and the
# cycles waves spend waiting for instructions.  Run on the GPU box through gpurun; writes under gpurun_out/r3/icache.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
rocprofv3 -L 2>/dev/null | grep -o "SQC\?_[A-Z_0-9]*\(ICACHE\|IFETCH\|INST_CACHE\|WAIT_INST\)[A-Z_0-9]*" | sort -u > $OUT/counter_names.txt
cat $OUT/counter_names.txt
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace -f csv -d $OUT/pmc_a -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-run-batch --repeats 5 > $OUT/a.json 2> $OUT/a.err || echo "pmc a failed"
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace -f csv -d $OUT/pmc_b -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-run-batch --repeats 5 > $OUT/b.json 2> $OUT/b.err || echo "pmc b failed"
python3 - <<'PY'
import csv, glob, collections, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r3/icache")
for d in ("pmc_a", "pmc_b"):
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if "ode_elbo" in k or "enc_" in k:
                print(d, k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})
PY
