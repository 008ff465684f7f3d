"""Deterministic synthetic batches shaped like the reference's three data sets.  No reference data is shipped;
these mirror the value ranges the reference's loaders produce: observations min-max normalised to [0,1]
(utils/ODE_dataset.py:196-209) with noise std 0.05 (data/cvs/cvs_data.py:16), Bernoulli labels for CVS/challenge, one-hot
devices + log1p concentrations and a slightly non-uniform time grid for proc (utils/proc_dataset.py:37-50,93)."""
from __future__ import annotations

from typing import Dict, Tuple

import torch


def synthetic_batch(family: str, B: int, T: int, C: int, seed: int = 1234) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
    """Returns (observations, labels, times).  observations is the reference layout: a [B,C,T] permuted view of a contiguous
    [B,T,C] tensor for cvs/challenge (training_cvs.py:25), contiguous [B,C,T] for proc (utils/proc_dataset.py:150)."""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(T, dtype=torch.float64)
    if family == "cvs":
        labels = {"iext": torch.bernoulli(torch.full((B, 1), 0.5), generator=g), "rtpr": torch.bernoulli(torch.full((B, 1), 0.5), generator=g)}
        times = torch.arange(0.0, T * 1.0, 1.0)
        drive = labels["iext"].double()
    elif family == "challenge":
        labels = {"symptoms": torch.bernoulli(torch.full((B, 1), 0.54), generator=g), "shedding": torch.bernoulli(torch.full((B, 1), 0.31), generator=g)}
        times = torch.arange(0.0, T * 1.0, 1.0)
        drive = labels["shedding"].double()
    elif family == "proc":
        aR = torch.nn.functional.one_hot(torch.randint(0, 3, (B,), generator=g), 3).float()
        aS = torch.nn.functional.one_hot(torch.randint(0, 4, (B,), generator=g), 4).float()
        c = torch.log1p(torch.rand(B, 2, generator=g) * 25000.0)
        labels = {"aR": aR, "aS": aS, "C12": c[:, :1].contiguous(), "C6": c[:, 1:].contiguous()}
        times = (0.1944 * torch.arange(T) + (torch.rand(T, generator=g) - 0.5) * 0.002).float()
        times[0] = 0.0
        drive = (c[:, :1] / 10.0).double()
    else:
        raise ValueError(family)
    base = torch.rand(B, C, 1, generator=g).double() * 0.2
    gain = 0.3 + 0.5 * torch.rand(B, C, 1, generator=g).double() + 0.2 * drive.unsqueeze(-1)
    tau = torch.tensor([10.0, 20.0, 40.0, 15.0][:C], dtype=torch.float64).reshape(1, C, 1) * (T / 86.0)
    curve = base + gain * (1 - torch.exp(-t.reshape(1, 1, T) / tau))
    obs = (curve + 0.05 * torch.randn(B, C, T, generator=g).double()).clamp(0, 1).float()
    if family != "proc":
        obs = obs.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    return obs, labels, times.float()
