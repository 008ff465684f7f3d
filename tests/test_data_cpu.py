"""Real-data front end: readers, transforms, splits and the pinned-buffer feeder, on files written by the test in the
reference's on-disk formats; when the reference tree is present (this container, not the GPU box) the readers are also run on its
shipped data files (shapes and label conventions only -- nothing from those files is copied)."""
import os
import pickle

import numpy as np
import pytest
import torch

from structured_latent_odes_amd import data as D

REF = "/root/reference/data"


def _write_cvs(tmp, n_train=40, n_test=7, T=30, C=3, seed=0):
    rng = np.random.default_rng(seed)
    obs = {"train": rng.normal(size=(n_train, T, C)) * 3 + 50, "test": rng.normal(size=(n_test, T, C)) * 3 + 50}
    par = lambda n: {"i_ext": rng.choice([0.0, -0.2], size=n), "r_tpr_mod": rng.choice([0.0, 0.5], size=n)}
    trp, tep = par(n_train), par(n_test)
    torch.save(obs, os.path.join(tmp, "processed_data.pkl"))
    torch.save(trp, os.path.join(tmp, "train_params_data.pkl"))
    torch.save(tep, os.path.join(tmp, "test_params_data.pkl"))
    return obs, trp, tep


def test_cvs_dataset_split_labels_and_transform(tmp_path):
    obs, trp, tep = _write_cvs(str(tmp_path))
    params = D.find_norm_params(obs["train"])
    assert np.allclose(params["std"], obs["train"].reshape(-1, 3).std(0)) and np.allclose(params["max"], obs["train"].max((0, 1)))
    tf = D.create_transforms("zero_to_one", params)
    tr = D.CVSDataset(str(tmp_path) + "/", "train", 20, False, tf)
    va = D.CVSDataset(str(tmp_path) + "/", "val", 20, False, tf)
    te = D.CVSDataset(str(tmp_path) + "/", "test", 20, False, tf)
    assert (len(tr), len(va), len(te)) == (36, 4, 7)                      # first 90 % / rest / test
    s = va[1]
    want = (obs["train"][37, :20] - params["min"]) / (params["max"] - params["min"])
    assert s["observations"].shape == (20, 3) and np.allclose(s["observations"].numpy(), want, atol=1e-6)
    assert s["iext"].item() == float(trp["i_ext"][37] >= 0) and s["rtpr"].item() == float(trp["r_tpr_mod"][37] > 0)
    back = tf["normalize"].denormalize(s["observations"][None])
    assert np.allclose(back[0].numpy(), obs["train"][37, :20], rtol=1e-5, atol=1e-3)
    z = D.create_transforms("zscore", params)["normalize"]
    assert np.allclose(z(torch.as_tensor(obs["train"][0])).numpy(), (obs["train"][0] - params["mean"]) / params["std"], atol=1e-5)
    with pytest.raises(ValueError):
        D.create_transforms("nope", params)
    # random window start stays inside the series
    rs = D.CVSDataset(str(tmp_path) + "/", "train", 20, True, tf)
    assert all(rs[i]["observations"].shape == (20, 3) for i in range(10))


def test_kfold_and_holdout_splits():
    tr, va = D.kfold_ids(35, 5, 5, 12)
    assert len(va) == 7 and len(tr) == 28 and not set(tr) & set(va) and sorted(set(tr) | set(va)) == list(range(35))
    assert np.array_equal(va, np.sort(va)) and np.array_equal(D.kfold_ids(35, 5, 5, 12)[1], va)          # seeded
    folds = [D.kfold_ids(35, 5, k, 12)[1] for k in range(1, 6)]
    assert sorted(np.concatenate(folds).tolist()) == list(range(35))                                      # folds partition the data
    np.random.seed(12)
    assert np.array_equal(va, np.sort(np.array_split(np.random.permutation(35), 5)[4]))                 # the reference's recipe
    tr, va = D.holdout_ids(np.array([0, 1, 2, 1, 0, 2, 2]), 2)
    assert va.tolist() == [2, 5, 6] and tr.tolist() == [0, 1, 3, 4]


def test_challenge_pair_and_feeder(tmp_path):
    rng = np.random.default_rng(1)
    data = {"observations": rng.random((35, 142, 4)), "shedding": rng.integers(0, 2, (35, 1)).astype(float),
            "symptoms": rng.integers(0, 2, (35, 1)).astype(float), "n_time": 142}
    path = str(tmp_path / "data.pkl")
    with open(path, "wb") as fh:
        pickle.dump(data, fh)
    pair = D.build_challenge_datasets(path, seed=12, folds=5, split=5)
    assert pair.n_train == 28 and pair.n_test == 7 and pair.max_time == 142
    assert np.allclose(pair.data_norm_params["min"], pair.train["observations"].min((0, 1)))
    ds = D.ChallengeDataset(pair.train, transforms=D.create_transforms("zero_to_one", pair.data_norm_params))
    feeder = D.BatchFeeder(ds, batch_size=10, device=torch.device("cpu"))
    batches = list(feeder)
    assert len(feeder) == 3 and [b["observations"].shape[0] for b in batches] == [10, 10, 8]
    assert batches[0]["observations"].shape == (10, 142, 4) and batches[0]["observations"].is_contiguous()      # [B, T, C]
    assert torch.equal(batches[2]["observations"][3], ds[23]["observations"]) and torch.equal(batches[1]["shedding"][0], ds[10]["shedding"])
    assert float(batches[0]["observations"].min()) >= -1e-6 and float(batches[0]["observations"].max()) <= 1.0 + 1e-6   # float32 rounding of the reference's own formula
    sh = D.BatchFeeder(ds, batch_size=10, device=torch.device("cpu"), shuffle=True, seed=3, drop_last=True)
    got = torch.cat([b["shedding"] for b in sh])
    assert got.shape[0] == 20 and len(sh) == 2


def test_proc_csv_reader(tmp_path):
    # two tiny plate-reader files in the reference's column layout
    def write(name, devs, conds, T, t_scale):
        sigs = ["EYFP", "ECFP", "mRFP1", "OD"]
        cols = ["Content", "Colony", "Well Col", "Well Row", "Content"] + ["Raw Data (%s) %d - x" % (s, k + 1) for s in sigs for k in range(T)]
        times = ["", "", "", "", ""] + [t_scale * k for _ in sigs for k in range(T)]
        rows = [times]
        for i, (d, c) in enumerate(zip(devs, conds)):
            rows.append([d, "", str(i + 1), "A", c] + [float(10 * si + i + 0.01 * k) for si in range(4) for k in range(T)])
        import csv
        with open(os.path.join(str(tmp_path), name), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(cols)
            w.writerows(rows)
    write("a.csv", ["R33S32_Y81C76", "Pcat_Y81C76", "XXX"], ["C6=25000", "C12=1.5", "C6=1"], 6, 0.2)
    write("b.csv", ["R33S32_Y81C76", "R33S32_Y81C76"], ["C6=0", "EtOH=2"], 8, 0.15)
    devices = ["Pcat_Y81C76", "R33S32_Y81C76"]
    dmap = {d: float(i) for i, d in enumerate(devices)}
    sig = ["OD", "mRFP1", "EYFP", "ECFP"]
    pa = D.load_proc_csv(str(tmp_path / "a.csv"), devices, dmap, ["C6", "C12"], sig)
    pb = D.load_proc_csv(str(tmp_path / "b.csv"), devices, dmap, ["C6", "C12"], sig)
    dev, tr, times, obs = pa
    assert dev.tolist() == [1, 0] and tr.tolist() == [[25000.0, 0.0], [0.0, 1.5]] and obs.shape == (2, 4, 6)
    assert np.allclose(times, 0.2 * np.arange(6)) and np.allclose(obs[0, 0], 30 + 0 + 0.01 * np.arange(6))     # signal order: OD first
    # the reference derives the list of "other" conditions from the FIRST kept row of a file (load_proc_data.py:18-26): b.csv's first
    # row only names C6, so its EtOH row stays (with zero C6 / C12); a file whose first row names EtOH would drop it
    assert pb[0].tolist() == [1, 1] and pb[1].tolist() == [[0.0, 0.0], [0.0, 0.0]] and pb[3].shape == (2, 4, 8)
    write("c.csv", ["R33S32_Y81C76", "R33S32_Y81C76", "Pcat_Y81C76"], ["EtOH=0;C6=3", "EtOH=2", "C12=1"], 8, 0.15)
    pc = D.load_proc_csv(str(tmp_path / "c.csv"), devices, dmap, ["C6", "C12"], sig)
    assert pc[0].tolist() == [1, 0] and pc[1].tolist() == [[3.0, 0.0], [0.0, 1.0]]                              # EtOH = 2 row dropped
    assert D.load_proc_csv(str(tmp_path / "a.csv"), ["nope"], {"nope": 0.0}, ["C6"], sig) is None
    ds = D.ProcDataset([pa, pb], subtract_background=True, dev_1hot_fn=lambda d: np.eye(2, dtype=np.float32)[d])
    assert len(ds) == 4 and ds.observations.shape == (4, 4, 6)                                              # grid of the file with fewest series
    assert np.allclose(ds.inputs[0].numpy(), np.log(1.0 + np.array([25000.0, 0.0])))
    assert float(ds.observations.min()) == 0.0 and float(ds.observations.max()) <= 1.0
    assert ds[2]["dev_1hot"].tolist() == [0.0, 1.0]
    # the training view of training_proc.py:23-32: aR | aS cassettes, C12 := inputs[:, 0], C6 := inputs[:, 1], observations [C, T]
    tr, va, times = D.build_proc_datasets(str(tmp_path), seed=12, folds=3, split=1, files=["a.csv", "b.csv"])
    assert len(tr) + len(va) == 4 and len(va) == 2 and times.shape == (6,)
    item = tr[0]
    assert item["observations"].shape == (4, 6) and item["aR"].shape == (3,) and item["aS"].shape == (4,)
    assert float(item["aR"].sum()) == 1.0 and float(item["aS"].sum()) == 1.0 and item["C12"].shape == (1,) and item["C6"].shape == (1,)
    assert D.proc_cassettes(np.array([0, 3, 5])).tolist() == [[1, 0, 0, 1, 0, 0, 0], [0, 0, 1, 0, 1, 0, 0], [0, 0, 1, 0, 0, 0, 1]]
    tr_h, va_h, _ = D.build_proc_datasets(str(tmp_path), seed=12, folds=3, split=1, heldout="Pcat_Y81C76", files=["a.csv", "b.csv"])
    assert len(va_h) == 1 and float(va_h[0]["aR"][0]) == 1.0 and len(tr_h) == 3


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference data tree not present (GPU box)")
def test_readers_on_the_shipped_reference_files():
    tr = D.CVSDataset(REF + "/cvs/", "train", 86, False, D.create_transforms("zero_to_one", D._torch_load(REF + "/cvs/data_norm_params.pkl")))
    assert len(tr) == 810 and tr[0]["observations"].shape == (86, 3) and set(float(tr[i]["iext"]) for i in range(50)) <= {0.0, 1.0}
    assert len(D.CVSDataset(REF + "/cvs/", "val", 86, False)) == 90 and len(D.CVSDataset(REF + "/cvs/", "test", 86, False)) == 100
    pair = D.build_challenge_datasets(REF + "/challenge/data.pkl", seed=12, folds=5, split=5)
    assert (pair.n_train, pair.n_test, pair.max_time) == (28, 7, 142) and pair.train["observations"].shape == (28, 142, 4)
    devices = ["Pcat_Y81C76", "RS100S32_Y81C76", "RS100S34_Y81C76", "R33S32_Y81C76", "R33S34_Y81C76", "R33S175_Y81C76"]
    dmap = {d: float(i) for i, d in enumerate(devices)}
    parsed = D.load_proc_csv(REF + "/proc/proc140916.csv", devices, dmap, ["C6", "C12"], ["OD", "mRFP1", "EYFP", "ECFP"])
    assert parsed is not None and parsed[3].shape[1:] == (4, 100) and parsed[1].shape[1] == 2 and len(parsed[2]) == 100
    tr, va, times = D.build_proc_datasets(REF + "/proc", seed=12, folds=4, split=1)
    assert len(tr) == 234 and len(va) == 78 and times.shape == (100,) and tr[0]["observations"].shape == (4, 100)
    assert 0.0 <= float(tr.ds.observations.min()) and float(tr.ds.observations.max()) <= 1.0


# ---- pinned against the reference's own loaders (tests/golden/g5_data.npz, written by tests/golden/make_golden_data.py) ----------
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g5_data.npz")
needs_ref_data = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "proc", "proc140916.csv")),
                                    reason="the reference's shipped data files are only present in the build container")


@needs_ref_data
def test_cvs_reader_equals_reference_loader():
    g = np.load(GOLD)
    import torch as _t
    params = _t.load(os.path.join(REF, "cvs", "data_norm_params.pkl"), weights_only=False)
    tf = D.create_transforms("zero_to_one", params)
    assert np.allclose(np.asarray(tf["normalize"].min_val, dtype=np.float64), g["cvs.norm.min"], rtol=1e-6)
    for split in ("train", "val", "test"):
        ds = D.CVSDataset(os.path.join(REF, "cvs") + "/", split, 86, False, tf)
        assert len(ds) == int(g["cvs.%s.len" % split])
        first = [ds[i] for i in range(3)] + [ds[len(ds) - 1]]
        assert np.array_equal(np.stack([s["observations"].numpy() for s in first]), g["cvs.%s.obs" % split])     # bit-exact
        assert [float(s["iext"]) for s in first] == list(g["cvs.%s.iext" % split])
        assert [float(s["rtpr"]) for s in first] == list(g["cvs.%s.rtpr" % split])
        sums = [float(sum(ds[i]["iext"].item() for i in range(len(ds)))), float(sum(ds[i]["rtpr"].item() for i in range(len(ds))))]
        assert sums == list(g["cvs.%s.label_sums" % split])


@needs_ref_data
def test_challenge_split_and_norm_equal_reference_loader():
    g = np.load(GOLD)
    pair = D.build_challenge_datasets(os.path.join(REF, "challenge", "data.pkl"), seed=12, folds=5, split=5)
    assert (pair.n_train, pair.n_test, pair.max_time) == (int(g["challenge.n_train"]), int(g["challenge.n_test"]), int(g["challenge.max_time"]))
    for k in ("mean", "std", "max", "min"):
        assert np.array_equal(np.asarray(pair.data_norm_params[k], dtype=np.float64), g["challenge.norm.%s" % k]), k
    for name, d in (("train", pair.train), ("test", pair.test)):
        want_sum = float(g["challenge.%s.obs_sum" % name])
        assert abs(np.asarray(d["observations"], dtype=np.float64).sum() - want_sum) <= 1e-12 * abs(want_sum)
        assert np.array_equal(np.asarray(d["shedding"], dtype=np.float64), g["challenge.%s.shedding" % name])
        assert np.array_equal(np.asarray(d["symptoms"], dtype=np.float64), g["challenge.%s.symptoms" % name])
    ds = D.ChallengeDataset(pair.test, "val", 142, False, D.create_transforms("zero_to_one", pair.data_norm_params))
    assert np.array_equal(np.stack([ds[i]["observations"].numpy() for i in range(3)]), g["challenge.val.obs"])
    got = D.find_norm_params(g["norm_params.synthetic_in"])
    for k in ("mean", "std", "max", "min"):
        assert np.allclose(got[k], g["norm_params.synthetic.%s" % k], rtol=1e-12, atol=0)


@needs_ref_data
def test_proc_front_end_equals_reference_loader():
    g = np.load(GOLD)
    dmap = {d: float(i) for i, d in enumerate(D.PROC_DEVICES)}
    parsed = []
    for f in D.PROC_FILES:
        dev, treat, times, obs = D.load_proc_csv(os.path.join(REF, "proc", f), D.PROC_DEVICES, dmap, D.PROC_CONDITIONS, D.PROC_SIGNALS)
        assert np.array_equal(dev, g["proc.file.%s.devices" % f])
        assert np.array_equal(treat, g["proc.file.%s.treatments" % f])
        assert np.array_equal(times, g["proc.file.%s.times" % f])
        assert list(obs.shape) == list(g["proc.file.%s.obs_shape" % f])
        want_sum = float(g["proc.file.%s.obs_sum" % f])                   # float64 checksum of float32 values: order-of-summation slack only
        assert abs(obs.astype(np.float64).sum() - want_sum) <= 1e-12 * abs(want_sum)
        assert np.array_equal(obs[:2], g["proc.file.%s.obs_first" % f])
        parsed.append((dev, treat, times, obs))
    ds = D.ProcDataset(parsed, normalize=None, subtract_background=True, dev_1hot_fn=D.proc_cassettes)
    assert np.array_equal(ds.devices, g["proc.devices"])
    assert np.array_equal(ds.inputs.numpy(), g["proc.inputs"])
    assert np.array_equal(ds.times.numpy(), g["proc.times"])
    assert np.array_equal(np.asarray(ds.scales, dtype=np.float64), g["proc.scales"])
    assert list(ds.observations.shape) == list(g["proc.obs_shape"])
    assert np.array_equal(ds.observations[:3].numpy(), g["proc.obs_first"]) and np.array_equal(ds.observations[-2:].numpy(), g["proc.obs_last"])
    assert abs(ds.observations.double().sum().item() - float(g["proc.obs_sum"])) <= 1e-9 * abs(float(g["proc.obs_sum"]))
    assert np.array_equal(ds.dev_1hot.numpy(), g["proc.dev_1hot"])
    for split in (1, 3):
        tr, va = D.kfold_ids(len(ds), 4, split, 12)
        assert np.array_equal(tr, g["proc.fold%d.train_ids" % split]) and np.array_equal(va, g["proc.fold%d.val_ids" % split])
    tr, va = D.holdout_ids(ds.devices, int(dmap["R33S34_Y81C76"]))
    assert np.array_equal(tr, g["proc.heldout.train_ids"]) and np.array_equal(va, g["proc.heldout.val_ids"])
    # the training view and builder used by training_proc.py
    tv, vv, times = D.build_proc_datasets(os.path.join(REF, "proc"), seed=12, folds=4, split=1)
    assert len(tv) == len(g["proc.fold1.train_ids"]) and len(vv) == len(g["proc.fold1.val_ids"])
    assert np.array_equal(times.numpy(), g["proc.times"].astype(np.float32))


def test_golden_data_file_is_data_only():
    g = np.load(GOLD)
    assert len(g.files) > 50 and all(g[k].dtype.kind in "fiub" for k in g.files)
