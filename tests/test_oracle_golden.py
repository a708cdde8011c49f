"""The CPU oracle (oracle/neuralbody_oracle.py) against the fixtures that
tests/golden/make_golden.py produced by running the UNMODIFIED reference.
Tolerances: the oracle and the reference are both fp32 CPU torch, so they agree
to fp32 round-off (<= 5e-6 on O(1) quantities)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import neuralbody_oracle as orc
from tests.golden import scenes

TOL = 5e-6


def _digest(sd, batch):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    for k in sorted(batch):
        h.update(k.encode())
        h.update(np.ascontiguousarray(batch[k]).tobytes())
    return h.hexdigest()


def _close(a, b, tol=TOL, name=""):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), name + ": NaN pattern differs"
    err = np.abs(a[~nan_a] - b[~nan_a])
    scale = np.maximum(1.0, np.abs(b[~nan_a]))
    assert (err / scale).max(initial=0.0) <= tol, (name, float((err / scale).max()))


def test_oracle_render_with_trained_weights_matches_reference_golden(golden_dir):
    """The fixture made by 300 Adam steps of the reference's NetworkWrapper (make_golden.py::run_trained): the oracle renders
    the same image from the stored parameters."""
    g = np.load(os.path.join(golden_dir, "scene_small_trained.npz"))
    params = {k[len("param/"):]: g[k] for k in g.files if k.startswith("param/")}
    assert sum(v.size for v in params.values()) > 500000
    r, sd, body, batch, cam, _ = scenes.build_trained(params)
    with torch.no_grad():
        out = orc.render(orc.tensor_state_dict(sd), batch, n_samples=r["n_samples"], training=True, white_bkgd=r["white_bkgd"])
    for k in ("rgb_map", "acc_map", "weights", "depth_map"):
        _close(out[k].numpy(), g[k], tol=2e-5, name=k)
    # the optimiser moved the decoder: its weights no longer look like the initialiser's
    base = scenes.build(scenes.TRAINED["base"])[1]
    assert float(np.abs(params["fc_1.weight"] - base["fc_1.weight"]).max()) > 0.02


@pytest.mark.parametrize("tag", ["512x64"])
def test_oracle_on_the_bench_scene_matches_the_reference_run(tag, golden_dir):
    """Parity AT SIZE held against the reference itself: 384 rays of the headline 512 x 512 x 64 view (pose 1
    of bench.py's timed cycle, full out_sh, train-mode BatchNorm) as the unmodified reference renders them
    (make_golden.py::run_bench) — the oracle that bench.py and tests/test_gpu_fullsize.py check every ray against reproduces
    them.  (The 1024 x 1024 x 128 fixture is the same scene and encoder through another camera: the GPU test reads it.)"""
    g = np.load(os.path.join(golden_dir, "bench_%s.npz" % tag))
    r, sd, body, batch, pick = scenes.build_bench(tag)
    assert _digest(sd, batch) == str(g["input_digest"]), "seeded inputs drifted from the fixture"
    assert np.array_equal(pick, g["pick"]) and len(pick) == scenes.N_BENCH_RAYS
    with torch.no_grad():
        out = orc.render(orc.tensor_state_dict(sd), batch, n_samples=r["n_samples"], training=True)
    for k in ("rgb_map", "disp_map", "acc_map", "weights", "depth_map"):
        _close(out[k].numpy(), g[k], tol=2e-5, name=k)
    raw = out["raw"][0].reshape(len(pick), r["n_samples"], 4).numpy()
    _close(raw[:, -1, 3], g["sigma_last"], tol=1e-4, name="sigma_last")  # two fp32 evaluations of a 256-term sum with |terms| ~ 10
    assert float(g["rgb_map"].max() - g["rgb_map"].min()) > 0.3 and float(np.abs(g["sigma_last"]).min()) > 1e-3


def test_oracle_frame_by_frame_matches_the_batch_of_two_fixture(golden_dir):
    """scenes.BATCH2 (B = 2): the reference renderer run on each frame at the batch's common out_sh (make_golden.run_batch2; the
    reference itself cannot run the B = 2 batch — the fixture records the exception it raises) against the oracle frame by frame."""
    g = np.load(os.path.join(golden_dir, "scene_batch2.npz"))
    r, sd, batch, frames = scenes.build_batch2()
    assert _digest(sd, batch) == str(g["input_digest"]), "seeded inputs drifted from the fixture"
    assert not bool(g["reference_runs_b2"])
    assert batch["ray_o"].shape == (2, r["n_rays"], 3) and not np.array_equal(batch["out_sh"][0], batch["out_sh"][1])
    sdt = orc.tensor_state_dict(sd)
    for b, f in enumerate(frames):
        assert np.array_equal(f["out_sh"][0], batch["out_sh"].max(0))
        with torch.no_grad():
            out = orc.render(dict(sdt), f, n_samples=r["n_samples"], training=True)
        for k in ("rgb_map", "acc_map", "weights", "depth_map"):
            _close(out[k].numpy(), g[k][b:b + 1], tol=2e-5, name="%s[%d]" % (k, b))
    assert float(g["acc_map"].mean()) > 0.1


@pytest.mark.parametrize("name", list(scenes.SCENES))
def test_oracle_render_matches_reference_golden(name, golden_dir):
    g = np.load(os.path.join(golden_dir, "scene_%s.npz" % name))
    r, sd, body, batch, cam, t_rand = scenes.build(name)
    assert _digest(sd, batch) == str(g["input_digest"]), "seeded inputs drifted from the fixture"
    sdt = orc.tensor_state_dict(sd)
    stats = {}
    with torch.no_grad():
        out_sh = batch["out_sh"].max(0).tolist()
        vols = orc.encode_sparse_voxels(sdt, torch.from_numpy(batch["coord"]), out_sh,
                                        training=(r["mode"] == "train"), update_stats=stats)
        out = orc.render(sdt, batch, n_samples=r["n_samples"], training=(r["mode"] == "train"),
                         t_rand=None if t_rand is None else torch.from_numpy(t_rand),
                         white_bkgd=r["white_bkgd"], feature_volume=vols)
    for k in ("rgb_map", "disp_map", "acc_map", "weights", "depth_map"):
        # disp = 1 / (depth / acc) divides two fp32 sums whose summation order differs with the chunk shape
        _close(out[k].numpy(), g[k], tol=2e-5 if k == "disp_map" else TOL, name=k)
    if "raw_subset" in g:
        ns = r["n_samples"]
        raw = out["raw"].view(1, -1, ns, 4)[:, ::scenes.RAW_RAY_STRIDE].reshape(1, -1, 4)
        _close(raw.numpy(), g["raw_subset"], tol=2e-4, name="raw")  # fp32 GEMM order differs with batch shape
    if r["probes"]:
        for li, v in enumerate(vols):
            flat = v[0].permute(1, 2, 3, 0).reshape(-1, v.shape[1]).numpy()
            assert list(v.shape) == list(g["vol%d_shape" % li])
            _close(flat[g["vol%d_probe_idx" % li]], g["vol%d_probe_val" % li], tol=1e-4, name="vol%d" % li)  # 17 fp32 conv+BN layers
            assert int((np.abs(flat).sum(1) > 0).sum()) == int(g["vol%d_nonzero_voxels" % li])
            assert abs(flat.astype(np.float64).sum() - float(g["vol%d_sum" % li])) <= 1e-5 * float(g["vol%d_abs_sum" % li])
    for k, v in stats.items():
        _close(v.numpy(), g["bn/" + k], tol=1e-5, name=k)


def test_oracle_raygen_matches_reference_golden(golden_dir):
    from tests import synthetic as syn

    g = np.load(os.path.join(golden_dir, "raygen.npz"))
    for tag, body_kw, H, W, ff in (("a", dict(seed=3, box=(0.9, 1.7, 0.35), rh=(0.2, 0.4, 0.0), th=(0.3, 0.1, 0.2)), 40, 56, 1.1),
                                   ("b", dict(seed=4, box=(0.3, 0.5, 0.2)), 33, 17, 3.0)):
        body = syn.make_body(**body_kw)
        K, R, T = syn.make_camera(body, H, W, focal_factor=ff, distance=2.2, yaw=-0.6, pitch=0.25)
        ro, rd, near, far, mask = orc.image_rays(H, W, K, R, T, body["can_bounds"])
        assert np.array_equal(mask, g[tag + "_mask"])
        assert 0 < mask.sum() < mask.size or tag == "b"
        np.testing.assert_array_equal(rd, g[tag + "_img_ray_d"])
        np.testing.assert_array_equal(near, g[tag + "_img_near"])
        np.testing.assert_array_equal(far, g[tag + "_img_far"])
        np.testing.assert_array_equal(ro[0], g[tag + "_ray_o"])


def test_oracle_autograd_matches_reference_training_step(golden_dir):
    """The oracle under autograd against the gradients the UNMODIFIED reference produced for one training step
    (tests/golden/train_step.npz, made by make_golden.py: NetworkWrapper on CPU): loss, and per parameter the
    gradient norm, sum and probe entries."""
    g = np.load(os.path.join(golden_dir, "train_step.npz"))
    r, sd, batch, t_rand = scenes.build_train()
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v)
           for k, v in orc.tensor_state_dict(sd).items()}
    out = orc.render(sdg, batch, n_samples=r["n_samples"], training=True, t_rand=torch.from_numpy(t_rand))
    mask = torch.from_numpy(batch["mask_at_box"])
    loss = torch.mean((out["rgb_map"][mask] - torch.from_numpy(batch["rgb"])[mask]) ** 2)  # if_nerf_clight.py:25
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= 1e-6
    _close(out["rgb_map"].detach().numpy(), g["rgb_map"], name="rgb_map")
    names = [k[5:] for k in g.files if k.startswith("norm/")]
    assert len(names) == 69  # 17 conv + 34 BN affine + 16 MLP + c + latent: every trainable tensor
    for name in names:
        gr = sdg[name].grad.numpy().astype(np.float64)
        scale = max(float(g["max/" + name]), 1e-30)
        assert abs(np.sqrt((gr ** 2).sum()) - float(g["norm/" + name])) <= 2e-3 * max(float(g["norm/" + name]), 1e-30), name
        idx = scenes.grad_probe_indices(gr.shape)
        assert np.abs(gr.reshape(-1)[idx] - g["probe/" + name]).max() <= 2e-3 * scale, name


@pytest.mark.parametrize("kind", ["mmsk", "msk"])
def test_oracle_masked_renderers_match_reference_golden(kind, golden_dir):
    """oracle.render_masked against the reference's if_clight_renderer_mmsk / _msk (fixtures made by make_golden.py)."""
    g = np.load(os.path.join(golden_dir, "masked_%s.npz" % kind))
    r, sd, batch, (Hh, Ww) = scenes.build_masked(kind)
    assert _digest(sd, batch) == str(g["input_digest"]), "seeded inputs drifted from the fixture"
    with torch.no_grad():
        out = orc.render_masked(orc.tensor_state_dict(sd), batch, Hh, Ww, kind, n_samples=r["n_samples"], training=True)
    assert np.array_equal(out["inside"].numpy(), g["inside"])
    assert 0.2 < g["inside"].mean() < 0.8, "fixture does not exercise the culling"
    for k in ("rgb_map", "disp_map", "acc_map", "weights", "depth_map"):
        _close(out[k].numpy(), g[k], tol=2e-5 if k == "disp_map" else TOL, name=k)


def test_oracle_density_cube_matches_reference_golden(golden_dir):
    """oracle.density_cube against the reference's if_mesh_renderer cube (fixture made by make_golden.py::run_mesh)."""
    g = np.load(os.path.join(golden_dir, "mesh_cube.npz"))
    r, sd, batch = scenes.build_mesh()
    assert _digest(sd, batch) == str(g["input_digest"]), "seeded inputs drifted from the fixture"
    with torch.no_grad():
        cube = orc.density_cube(orc.tensor_state_dict(sd), batch, training=True)
    assert cube.shape == g["cube"].shape and cube.dtype == np.float64
    assert int((cube != 0).sum()) <= int(g["n_inside"])
    assert (g["cube"] > 5).sum() > 100 and (g["cube"] < 5).sum() > 100, "fixture does not straddle cfg.mesh_th"
    _close(cube, g["cube"], tol=5e-5, name="cube")  # alpha_fc is scaled x12 in the recipe: fp32 summation-order noise
