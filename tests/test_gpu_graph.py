"""The training step inside a hipGraph (tools/graph_probe.py): every library call of a step -- main and auxiliary -- can be
captured on the caller's stream (no allocation, host synchronisation or host read-back inside a step), and ONE replay of a block of K
captured steps is bitwise the K stream-launched steps from the same weights and Adam state (same kernels, same arguments; the Adam
launch's bias corrections are kernel arguments, so a SECOND replay would repeat step counts: that is the part a product form still
needs, and this test replays once)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _case(fam, T, B, kw):
    import importlib
    from structured_latent_odes_amd import configs as CF
    from structured_latent_odes_amd.svi import AuxStep, ELBOStep, FlatAdam
    from structured_latent_odes_amd.synthetic import synthetic_batch
    from structured_latent_odes_amd.utils.utils import set_seed
    cfg = getattr(CF, "load_config_" + fam)(); cfg.update(seq_len=T, **kw)
    set_seed(5)
    mod = importlib.import_module("structured_latent_odes_amd.models.mechanistic_" + fam)
    obs, labels, times = synthetic_batch(fam, B, T, cfg.obs_dim)
    m = mod.MechanisticModel(cfg, DEV, times.to(DEV)); b = m._bind()
    opt = FlatAdam(b.engine, b.flat, lr=1e-3)
    labels_d = {k: v.to(DEV) for k, v in labels.items()}
    obs_d, u_d = obs.to(DEV), m.labels_to_u(**labels_d)
    eps_d = torch.randn(B, m.latent_dim, generator=torch.Generator().manual_seed(9)).to(DEV)
    main, aux = ELBOStep(b.engine, b.flat, opt), AuxStep(m, opt)

    def minibatch():   # training_cvs.py:147-157: both SVI objects, Adam after each
        main.step_async(obs_d, eps=eps_d, u=u_d)
        aux.step_async(obs_d, eps=eps_d, **labels_d)
    return b.flat, opt, main, aux, minibatch


@pytest.mark.parametrize("fam,T,B,kw", [("cvs", 200, 64, dict(z_iext_dim=3, z_rtpr_dim=3, z_epsilon_dim=2, solver="rk4")),
                                        ("cvs", 100, 32, dict(z_iext_dim=1, z_rtpr_dim=1, z_epsilon_dim=2, solver="midpoint")),
                                        ("proc", 100, 48, dict(solver="dopri5"))], ids=["metric_shape", "config0_midpoint", "proc_dopri5"])
def test_one_replay_of_captured_minibatches_is_the_stream_launched_run(fam, T, B, kw):
    K = 3
    flat, opt, main, aux, minibatch = _case(fam, T, B, kw)
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):          # workspaces and per-shape set-up happen on first use: outside the capture
        for _ in range(2):
            minibatch()
    torch.cuda.current_stream(DEV).wait_stream(side)
    torch.cuda.synchronize(DEV)
    snap = (flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.t)

    for _ in range(K):
        minibatch()
    torch.cuda.synchronize(DEV)
    want = (flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), main.loss.clone(), aux.loss.clone())

    flat.copy_(snap[0]); opt.exp_avg.copy_(snap[1]); opt.exp_avg_sq.copy_(snap[2]); opt.t = snap[3]
    torch.cuda.synchronize(DEV)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(K):
            minibatch()
    torch.cuda.synchronize(DEV)
    assert torch.equal(flat, snap[0]), "capturing must not execute anything"
    g.replay()
    torch.cuda.synchronize(DEV)
    got = (flat, opt.exp_avg, opt.exp_avg_sq, main.loss, aux.loss)
    for name, a, b in zip(("weights", "exp_avg", "exp_avg_sq", "main loss", "aux loss"), got, want):
        assert torch.isfinite(a).all(), name
        assert torch.equal(a, b), (name, (a - b).abs().max().item())
