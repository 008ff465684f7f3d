#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
TAG=${1:-x}
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/tfull_$TAG.log 2>&1; echo "tests rc=$?"; tail -16 $OUT/tfull_$TAG.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/benchfull_$TAG.json 2> $OUT/benchfull_$TAG.err; echo "bench rc=$?"
python - $TAG <<'PY'
import json,sys
d=json.load(open("gpurun_out/r3/benchfull_%s.json" % sys.argv[1]))
print("value %.3fM  ms %.5f" % (d["value"]/1e6, d["ms_per_step"]))
print({k: round(v,2) for k,v in d["roofline"]["kernel_us"].items()})
for o in d["other_configs"]:
    print(o.get("config"), "ms %.4f" % o.get("ms_per_step", -1), "frac %.3f" % o.get("frac", -1), {k: round(v,1) for k,v in o.get("kernel_us", {}).items()}, o.get("error", ""))
PY
STAMPS_CFG=c2 SLODE_LIB_PATH=structured_latent_odes_amd/libslode_stamps.so timeout -k 10 120 python tools/stamps.py > $OUT/stamps_c2_$TAG.log 2>&1; echo "stamps rc=$?"
head -24 $OUT/stamps_c2_$TAG.log
