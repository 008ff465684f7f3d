#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
TAG=${1:-x}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_gpu_instantiations.py -x -q -k "not dopri5" > $OUT/t_$TAG.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/t_$TAG.log
timeout -k 10 300 python bench.py --no-other-configs --no-cpu-baseline > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"
python - $TAG <<'PY'
import json,sys
d=json.load(open("gpurun_out/r3/bench_%s.json" % sys.argv[1]))
print("value %.3fM  ms %.5f" % (d["value"]/1e6, d["ms_per_step"]))
print({k: round(v,2) for k,v in d["roofline"]["kernel_us"].items()})
rb=d["run_batch"]; print({k: (round(v,5) if isinstance(v,float) else v) for k,v in rb.items() if k!="note" and k!="aux_kernel_us"}); print({k: round(v,2) for k,v in rb["aux_kernel_us"].items()})
PY
SLODE_LIB_PATH=structured_latent_odes_amd/libslode_stamps.so timeout -k 10 120 python tools/stamps.py > $OUT/stamps_$TAG.log 2>&1; echo "stamps rc=$?"
grep -E "setup|total|chain|weff|enc_fwd2|all 1024" $OUT/stamps_$TAG.log
