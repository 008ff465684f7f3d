"""Prints the fields of a bench.py JSON line that a round's A/B reading needs (argv[1] = file holding the line)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.0f traj/s  ms_per_step %.5f  frac %.4f  step_frac_fp32 %.4f" % (d["value"], d["ms_per_step"], r["frac"], r["step_frac_fp32"]))
print("kernel_us", {k: round(v, 2) for k, v in r["kernel_us"].items()}, "tail_us %.2f" % r["tail_us"])
print("frac_by_flop_count", {k: round(v["frac"], 4) for k, v in r.get("frac_by_flop_count", {}).items()})
print("sustained", d.get("sustained"))
print({k: v for k, v in d.items() if k.startswith("value_") or k.startswith("ms_per_step_")})
rb = d.get("run_batch") or {}
print("run_batch", {k: round(rb[k], 5) for k in rb if k.endswith("_ms")}, {k: round(rb[k]) for k in rb if k.startswith("traj")})
for o in d.get("other_configs", []):
    print(o.get("config"), o.get("ms_per_step"), {k: round(v, 1) for k, v in (o.get("kernel_us") or {}).items()}, o.get("error"))
for k in ("collective_us", "collective", "strong_scaling", "nccl_version"):
    if k in d:
        print(k, d[k])
print("cpu_baseline", d.get("cpu_baseline"))
