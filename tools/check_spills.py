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
    # The Makefile passes the list it expects (one .res per source of libslode.so): a missing one fails the gate, a stale one left
    # behind by a renamed source is not looked at.  Without arguments (the CPU test): the sources named by the Makefile's SRCS line.
    names = sys.argv[1:]
    if not names:
        srcs = re.search(r"^SRCS\s*:=\s*(.+)$", open(os.path.join(root, "Makefile")).read(), re.M).group(1).split()
        names = [s[:-4] + ".res" for s in srcs]
    files = [os.path.join(root, n) for n in names]
    missing = [n for n, f in zip(names, files) if not os.path.exists(f)]
    if missing:
        sys.exit("check_spills: no resource remarks for %s (build with the Makefile: each object is compiled together with its .res)" % ", ".join(missing))
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
