"""Diagnostic: run the dopri5 ELBO step repeatedly on the same inputs (`python tools/dp5_repro.py cvs|proc [B]`) and report which
gradient tensors / which entries of the reverse sweep's slab rows differ between runs (expected: none)."""
import sys, torch
sys.path.insert(0, ".")
from oracle import slode_oracle as O
from structured_latent_odes_amd import engine as E
fam = sys.argv[1]
dev = torch.device("cuda:0")
kw = dict(z_g=3, z_eps=2) if fam == "proc" else dict(z_iext=3, z_rtpr=3, z_eps=2)
S, T, B = (8, 100, 70) if fam == "proc" else (5, 60, 70)
B = int(sys.argv[2]) if len(sys.argv) > 2 else B
NR = (B + 31) // 32
mk_o, mk_e = (O.proc_spec, E.proc_spec) if fam == "proc" else (O.cvs_spec, E.cvs_spec)
ospec = mk_o(solver="dopri5", **kw)
espec = mk_e(solver="dopri5", **kw); espec.rtol, espec.atol = 1e-6, 1e-8
p = O.init_params(ospec, T=T, S=S)
g = torch.Generator().manual_seed(31)
p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
obs, u, eps, times = O.synthetic_batch(ospec, B, T)
if fam == "cvs": times = times * 0.25
eng = E.Engine(espec, T, dev); eng.set_times(times); flat = eng.pack(p)
obs_d = obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)
outs = []
for i in range(6):
    loss = torch.zeros(1, device=dev); grads = torch.zeros(eng.n_params, device=dev); x = torch.empty(B, T, S, device=dev)
    eng.elbo_step(flat, obs_d, u.to(dev), eps.to(dev), loss, grads=grads, x_out=x)
    torch.cuda.synchronize()
    # this kernel's slab rows inside the workspace (carve order of slode_api.hip; every block aligned to 64 floats)
    al = lambda n: (n + 63) & ~63
    sp = eng.spec
    FQ = sp.n_filters * (T - sp.filter_size + 1 - sp.pool_size + 1)
    L = sp.latent_dim
    off = 2 * al(B * L) + al(B * FQ) + al(B * sp.cnn_hidden_dim) + 2 * al(B * L) + al(B * 64)
    lay = eng.layout
    stride = al(lay.ode_end - lay.ode_begin + 1)
    ws = eng.workspace(B).view(torch.float32)
    rows = ws[off + B * stride: off + (B + NR) * stride].view(NR, stride).clone()
    outs.append((loss.item(), eng.unpack(grads), x.clone(), rows))
for i in range(1, 6):
    d = {k: (outs[i][1][k] - outs[0][1][k]).abs().max().item() for k in outs[0][1]}
    for r in range(NR):
        dr = (outs[i][3][r] - outs[0][3][r]).abs()
        nz = dr.nonzero().flatten().tolist()
        if nz: print("  row", r, "differing slab entries (index - 1 - o_wh):", [(n - 1 - (eng.layout.dyn_wh - eng.layout.ode_begin), round(float(dr[n]), 3)) for n in nz][:10], "count", len(nz))
    k0 = "decoder.ode_model.dynamics.dynamics_hidden.weight"
    dd = (outs[i][1][k0] - outs[0][1][k0]).abs()
    print("differing entries (row, col, diff):", [(int(r), int(c), round(float(dd[r, c]), 3)) for r, c in dd.nonzero()][:12])
    print("run", i, "loss equal", outs[i][0] == outs[0][0], "x equal", torch.equal(outs[i][2], outs[0][2]), {k: v for k, v in d.items() if v > 0})
