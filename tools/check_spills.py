#!/usr/bin/env python3
"""Build gate: no kernel of libslode.so may spill a VGPR.

hipcc (ROCm 7.2) can place the store of a live-through VGPR spill at the top of a control-flow join block, AHEAD of the `s_or_b64 exec`
that re-enables the lanes which skipped the branch: those lanes never store, and the later reload hands them stale scratch (round 1:
the persistent-loop T=300 instantiation of ode_elbo_kernel reloaded `chunk = tid >> 5` as 0 -- DESIGN.md 3.1).  SGPR spills go through
v_writelane / v_readlane, which ignore EXEC, and are only a speed matter.  Reads the `-Rpass-analysis=kernel-resource-usage` remarks
the Makefile saves next to every object (csrc/*.res).
"""
import glob
import os
import re
import sys


def kernels(path):
    txt = open(path).read()
    for blk in txt.split("Function Name: ")[1:]:
        name = blk.split()[0]
        f = lambda key: int(re.search(key + r": (\d+)", blk).group(1))
        yield name, f(r"VGPRs Spill"), f(r"SGPRs Spill"), f(r" VGPRs"), f(r"ScratchSize \[bytes/lane\]")


def main():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "structured_latent_odes_amd", "csrc")
    files = sorted(glob.glob(os.path.join(root, "*.res")))
    if not files:
        sys.exit("check_spills: no csrc/*.res files (build with the Makefile first)")
    bad, n = [], 0
    for path in files:
        for name, vsp, ssp, vg, scratch in kernels(path):
            n += 1
            if vsp:
                bad.append((os.path.basename(path), name, vsp, scratch))
    for b in bad:
        print("check_spills: %s: %s spills %d VGPRs (%d B scratch per lane)" % b, file=sys.stderr)
    if bad:
        sys.exit(1)
    print("check_spills: %d kernels in %d objects, no VGPR spills" % (n, len(files)))


if __name__ == "__main__":
    main()
