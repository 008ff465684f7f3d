"""Randomised parity sweep on the GPU (diagnostic; a fixed-seed slice of it runs in tests/test_gpu_fuzz.py).
Usage: python tools/fuzz_parity.py [n_cases] [seed]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_util import run_case

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
fails = 0
for case in range(n_cases):
    ok, desc = run_case(rng, case, dev)
    fails += 0 if ok else 1
    print(desc, "" if ok else "FAIL", flush=True)
print("cases %d  failures %d" % (n_cases, fails))
sys.exit(1 if fails else 0)
