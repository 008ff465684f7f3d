"""Binds the parameters of a module tree to the engine's flat parameter vector."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from ..engine import Engine, ModelSpec


class Binding:
    """Owns the Engine and the flat parameter vector; re-points every hot-path ``nn.Parameter`` at a view of it, and appends
    the parameters the kernels do not know (auxiliary classifier MLPs) after ``engine.n_params`` so one Adam covers all."""

    def __init__(self, spec: ModelSpec, times: torch.Tensor, device, named: Dict[str, nn.Parameter], extra: Optional[List[nn.Parameter]] = None):
        self.engine = Engine(spec, int(times.numel()), device)
        self.engine.set_times(times)
        eng = self.engine
        # the in-kernel noise generator takes its key from torch's global seed (utils.set_seed / torch.manual_seed: the reference seeds
        # everything through set_seed(config.seed), training_cvs.py:203) -- runs with the same seed draw the same noise
        eng.rng_seed(torch.initial_seed())
        extra = list(extra or [])
        n_extra = sum(p.numel() for p in extra)
        self.flat = torch.zeros(eng.n_params + n_extra, dtype=torch.float32, device=eng.device)
        self.slices: Dict[str, slice] = {}
        for key, off, shp in eng.param_table():
            p = named[key]
            n = p.numel()
            if tuple(p.shape) != tuple(shp):
                raise ValueError("parameter %s has shape %s, layout expects %s" % (key, tuple(p.shape), shp))
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(*shp)
            self.slices[key] = slice(off, off + n)
        off = eng.n_params
        self.extra = extra
        for p in extra:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1).to(eng.device))
            p.data = self.flat[off:off + n].view(p.shape)
            off += n
        self.named = named

    @property
    def n_total(self) -> int:
        return self.flat.numel()

    def grad_views(self, gbuf: torch.Tensor):
        """Point ``.grad`` of every bound parameter at the matching view of ``gbuf`` (autograd then accumulates in place)."""
        for key, p in self.named.items():
            sl = self.slices.get(key)
            if sl is not None:
                p.grad = gbuf[sl].view(p.shape)
        off = self.engine.n_params
        for p in self.extra:
            p.grad = gbuf[off:off + p.numel()].view(p.shape)
            off += p.numel()
