"""A fixed-seed slice of the randomised parity sweep (tools/fuzz_parity.py): shapes nobody curated -- small / odd T, every family,
solver, likelihood and gradient mode -- HIP ELBO step vs the fp64 oracle.  (It caught an LDS sizing bug for C*T < 128.)"""
import random

import pytest
import torch

from fuzz_util import run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [7, 11])
def test_randomised_shapes(seed):
    rng = random.Random(seed)
    dev = torch.device("cuda:0")
    bad = []
    for case in range(12):
        ok, desc = run_case(rng, case, dev)
        if not ok:
            bad.append(desc)
    assert not bad, bad
