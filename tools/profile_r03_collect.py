"""Copies the summaries of tools/profile_r03.sh (gpurun_out/r3/prof) into profiles/ under round-3 names and derives the two small
JSON files bench.py reads (traffic and instruction counts of the dominant kernel), both keyed to the sha1 of csrc/ode_kernel.hip
so that a stale figure is never attached to a changed kernel.
Usage: python tools/profile_r03_collect.py [gpurun_out/r3/prof] [tag]"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r3", "prof")
tag = sys.argv[2] if len(sys.argv) > 2 else "r03_a"
dst = os.environ.get("PROFILE_DST") or os.path.join(ROOT, "profiles")   # (on the GPU box: a directory under gpurun_out/, which is what travels back)
os.makedirs(dst, exist_ok=True)
sha = hashlib.sha1(open(os.path.join(ROOT, "structured_latent_odes_amd", "csrc", "ode_kernel.hip"), "rb").read()).hexdigest()
KERNELS = {"weff_kernel": "fold", "enc_fwd2_kernel": "enc_fwd", "ode_elbo_kernel": "ode_elbo", "enc_bwd_lin_kernel": "gemm", "enc_chain_kernel": "chain"}
ARMS = {0: "product: piecewise-linear heads + switching-sum contraction", 1: "direct head evaluation (v_fma, SGPR operands) + switching-sum contraction",
        2: "direct head evaluation + MFMA contraction (v_mfma_f32_16x16x4_f32)"}


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return name.split("(")[0].split("<")[0].replace(",", ";")[-60:]


def newest(files):
    """the scratch directory may hold several collections of the same command: only the latest one counts"""
    return [max(files, key=os.path.getmtime)] if files else []


def stats(d):
    rows = []
    for f in newest(glob.glob(os.path.join(d, "*", "*kernel_stats.csv"))):
        for r in csv.DictReader(open(f)):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"]), float(r["Percentage"])))
    return rows


def pmc(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest(glob.glob(os.path.join(d, "*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v[len(v) // 4:]) / len(v[len(v) // 4:]), len(v)) for c, v in cs.items()} for k, cs in out.items()}


def write_stats(rows, name):
    with open(os.path.join(dst, name), "w") as fh:
        fh.write("kernel,calls,avg_ns,min_ns,max_ns,percent\n")
        for r in sorted(rows, key=lambda r: -r[1] * r[2]):
            fh.write("%s,%d,%.1f,%.0f,%.0f,%.2f\n" % r)


def stats_full(d):
    """full kernel names (template arguments tell the shapes apart)"""
    rows = []
    for f in newest(glob.glob(os.path.join(d, "*", "*kernel_stats.csv"))):
        for r in csv.DictReader(open(f)):
            nm = r["Name"].replace("(anonymous namespace)::", "").replace(",", ";")[:110]
            rows.append((nm, int(r["Calls"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"]), float(r["Percentage"])))
    return rows


ab = {"source_sha1_ode_kernel_hip": sha, "workload": "bench.py --no-other-configs --no-run-batch: B=1024, T=200, rk4, fp32 (BASELINE config[1])", "arms": {}}
rows = stats(os.path.join(src, "stats_main"))
if rows:
    write_stats(rows, "%s_main_kernel_stats.csv" % tag)
    bj = os.path.join(src, "bench_main.json")
    if os.path.exists(bj):
        shutil.copy(bj, os.path.join(dst, "%s_main_bench_under_rocprof.json" % tag))
    ode = [r for r in rows if r[0] == "ode_elbo_kernel"]
    arm = {"what": ARMS[0], "ode_elbo_avg_us": round(ode[0][2] / 1e3, 2) if ode else None, "ode_elbo_calls": ode[0][1] if ode else None}
    for kind in ("sq", "mfma"):
        p = pmc(os.path.join(src, "pmc_%s" % kind)).get("ode_elbo_kernel")
        if p:
            arm["pmc_" + kind] = {c: round(v[0], 1) for c, v in p.items()}
    ab["arms"]["alg0"] = arm
json.dump(ab, open(os.path.join(dst, "%s_ode_elbo_ab.json" % tag), "w"), indent=1)
for sub, name in (("stats_full", "full_bench"), ("stats_config0", "config0"), ("stats_rk4fixed", "config2_rk4"), ("stats_dopri5", "config2_dopri5"),
                  ("stats_config4", "config4")):
    rows = stats_full(os.path.join(src, sub))
    if rows:
        write_stats(rows, "%s_%s_kernel_stats.csv" % (tag, name))
for f in ("bench_full.json",):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, "%s_%s" % (tag, f.replace("bench_full", "full_bench_under_rocprof"))))
for f, name in (("bench_config0.jsonl", "config0"), ("bench_rk4fixed.jsonl", "config2_rk4"), ("bench_dopri5.jsonl", "config2_dopri5"), ("bench_config4.jsonl", "config4")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, "%s_%s_bench_under_rocprof.jsonl" % (tag, name)))

fetch, write = pmc(os.path.join(src, "pmc_fetch")), pmc(os.path.join(src, "pmc_write"))
traffic = {"source_sha1_ode_kernel_hip": sha,
           "_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py (B=1024, T=200); hbm_bytes = "
                    "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md, HBM section); access widths "
                    "other than 16 B/lane are uncalibrated"}
for k, dom in KERNELS.items():
    if k in fetch and k in write and "FETCH_SIZE" in fetch[k] and "WRITE_SIZE" in write[k]:
        f, w = fetch[k]["FETCH_SIZE"][0], write[k]["WRITE_SIZE"][0]
        traffic[dom] = {"FETCH_SIZE_KB_avg": round(f, 1), "WRITE_SIZE_KB_avg": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(traffic, open(os.path.join(dst, "%s_pmc_traffic.json" % tag), "w"), indent=1)
for kind in ("sq", "mfma"):
    p = pmc(os.path.join(src, "pmc_%s" % kind))
    if p:
        with open(os.path.join(dst, "%s_pmc_%s.csv" % (tag, kind)), "w") as fh:
            fh.write("kernel,counter,launches,avg_value\n")
            for kn, cs in sorted(p.items()):
                for c, (v, n) in sorted(cs.items()):
                    fh.write("%s,%s,%d,%.1f\n" % (kn, c, n, v))
print(json.dumps(ab, indent=1)[:1500])
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in traffic.items() if isinstance(v, dict)}))
