"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py into profiles/<name>_pmc_traffic.json.
Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Each dir holds *_counter_collection.csv of `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -f csv -- python3 bench.py ...`.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import sys

KERNELS = {"weff_kernel": "fold", "enc_fwd2_kernel": "enc_fwd", "ode_elbo_kernel": "ode_elbo", "enc_bwd_lin_kernel": "gemm",
           "enc_chain_kernel": "chain"}


def collect(d, counter):
    out = collections.defaultdict(list)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            for key, dom in KERNELS.items():
                if key in r["Kernel_Name"]:
                    out[dom].append(float(r["Counter_Value"]))
    return out


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
res = {}
for dom in KERNELS.values():
    f, w = fetch.get(dom, []), write.get(dom, [])
    if not f or not w:
        continue
    f, w = f[len(f) // 4:], w[len(w) // 4:]   # drop warm-up launches
    fa, wa = sum(f) / len(f), sum(w) / len(w)
    res[dom] = {"FETCH_SIZE_KB_avg": round(fa, 1), "WRITE_SIZE_KB_avg": round(wa, 1), "launches": len(f),
                "hbm_bytes_per_launch": int((2 * fa + wa) * 1024)}
res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `bench.py --steps 20 --warmup 5` (B=1024, T=200, "
                "folded encoder path); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md "
                "section HBM); access widths other than 16 B/lane are uncalibrated, treat as +-2x")
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in res.items() if isinstance(v, dict)}))
