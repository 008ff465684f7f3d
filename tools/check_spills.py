#!/usr/bin/env python3
"""Build gate: no kernel of libslode.so may spill a VGPR -- to scratch or to the AGPR file.

hipcc (ROCm 7.2) can place the store of a live-through VGPR spill at the top of a control-flow join block, AHEAD of the `s_or_b64 exec`
that re-enables the lanes which skipped the branch: those lanes never store, and the later reload hands them stale scratch (round 1:
the persistent-loop T=300 instantiation of ode_elbo_kernel reloaded `chunk = tid >> 5` as 0 -- DESIGN.md 3.1).  SGPR spills go through
v_writelane / v_readlane, which ignore EXEC, and are only a speed matter.  A kernel whose block size lets it own more than 256 registers
per lane gets the overflow parked in AGPRs through v_accvgpr_write / read -- VALU moves that execute under EXEC exactly like a scratch
store, i.e. the same hazard class (round 2: the one-wave dopri5 reverse sweep, 256 VGPRs + 84..126 AGPRs, computed wrong dynamics
gradients with interval skipping enabled and right ones with any diagnostic compiled in) -- so AGPR use is rejected too, except in the
kernels listed in MFMA_KERNELS, whose AGPRs are matrix-core accumulators by design.  Reads the `-Rpass-analysis=kernel-resource-usage` remarks
the Makefile saves next to every object (csrc/*.res).
"""
import glob
import os
import re
import sys


MFMA_KERNELS = ("enc_bwd_lin_kernel",)


def kernels(path):
    txt = open(path).read()
    for blk in txt.split("Function Name: ")[1:]:
        name = blk.split()[0]
        f = lambda key: int(re.search(key + r": (\d+)", blk).group(1))
        yield name, f(r"VGPRs Spill"), f(r"SGPRs Spill"), f(r" VGPRs"), f(r"ScratchSize \[bytes/lane\]"), f(r"AGPRs")


def main():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "structured_latent_odes_amd", "csrc")
    files = sorted(glob.glob(os.path.join(root, "*.res")))
    if not files:
        sys.exit("check_spills: no csrc/*.res files (build with the Makefile first)")
    bad, n = [], 0
    for path in files:
        for name, vsp, ssp, vg, scratch, ag in kernels(path):
            n += 1
            if ag and not any(m in name for m in MFMA_KERNELS):
                vsp = max(vsp, ag)
            if vsp:
                bad.append((os.path.basename(path), name, vsp, scratch, ag))
    for b in bad:
        print("check_spills: %s: %s spills %d VGPRs (%d B scratch per lane, %d AGPRs)" % b, file=sys.stderr)
    if bad:
        sys.exit(1)
    print("check_spills: %d kernels in %d objects, no VGPR spills, no AGPR parking" % (n, len(files)))


if __name__ == "__main__":
    main()
