#!/bin/bash
# PMC counters of the config[2] dopri5 step's kernels (gpurun; one counter set per rocprofv3 run, kernel-trace only); summary printed per kernel name.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4/pmc_c2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -f csv -d $OUT/sq -- python3 tools/bench_configs.py "dopri5 (rtol" > $OUT/sq.log 2>&1 || echo "sq failed"
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace -f csv -d $OUT/mem -- python3 tools/bench_configs.py "dopri5 (rtol" > $OUT/mem.log 2>&1 || echo "mem failed"
python3 - <<'PY'
import csv, glob, collections, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r4/pmc_c2")
for sub in ("sq", "mem"):
    fs = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in fs:
        for r in csv.DictReader(open(f)):
            full = r["Kernel_Name"]
            kn = next((x for x in ("ode_elbo_kernel", "dopri5_kernel", "dopri5_lpt_kernel", "dopri5_bwd_kernel", "enc_fwd2_kernel", "weff_kernel", "enc_bwd_lin", "gemm", "enc_chain_kernel") if x in full), full[:40])
            acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", sub)
    for kn, d in acc.items():
        if not any(x in kn for x in ("ode_elbo", "dopri5", "enc_", "weff", "gemm", "chain")): continue
        print(kn, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
rm -rf $OUT/sq $OUT/mem
