#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_gpu_distributed.py -x -q -k "not dopri5" --durations=8 > $OUT/t2.log 2>&1; echo "tests rc=$?"; tail -12 $OUT/t2.log
timeout -k 10 300 python bench.py --no-other-configs --no-cpu-baseline > $OUT/bench2.json 2> $OUT/bench2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3/bench2.json"))
print("value %.3fM  ms %.5f" % (d["value"]/1e6, d["ms_per_step"]))
print({k: round(v,2) for k,v in d["roofline"]["kernel_us"].items()})
rb=d["run_batch"]; print({k: (round(v,5) if isinstance(v,float) else v) for k,v in rb.items() if k!="note" and k!="aux_kernel_us"}); print({k: round(v,2) for k,v in rb["aux_kernel_us"].items()})
PY
