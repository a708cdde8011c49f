"""Shared builders for the GPU parity tests: golden scenes -> device batch / Network / oracle."""
import os

import numpy as np
import torch

from tests.golden import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RGB_TOL = 5e-5  # BASELINE.json north_star asks <= 1e-4 RGB L-inf against the reference renderer; every arithmetic is held to half of it
# (measured worst case over the fixtures: 1.3e-5)
from neuralbody_amd.network import DEFAULT_PRECISION  # noqa: E402,F401  (the arithmetic Renderer.render uses by default)


def golden(name):
    return np.load(os.path.join(GOLDEN, "scene_%s.npz" % name))


def device_batch(batch_np, dev="cuda:0"):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in batch_np.items()}


def make_network(sd_np, dev="cuda:0", train=True, precision="f32"):
    from neuralbody_amd.network import Network

    net = Network(num_train_frame=sd_np["latent.weight"].shape[0], precision=precision)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    net = net.to(dev)
    net.train(train)
    return net


def make_renderer(net, recipe):
    from neuralbody_amd.renderer import RenderConfig, Renderer

    cfg = RenderConfig(N_samples=recipe["n_samples"], perturb=1.0 if recipe["perturb"] else 0.0,
                       raw_noise_std=0.0, white_bkgd=recipe["white_bkgd"])
    return Renderer(net, cfg)


def oracle_volumes(sd_np, batch_np, training):
    from oracle import neuralbody_oracle as orc

    sdt = orc.tensor_state_dict(sd_np)
    with torch.no_grad():
        out_sh = batch_np["out_sh"].max(0).tolist()
        vols = orc.encode_sparse_voxels(sdt, torch.from_numpy(batch_np["coord"]), out_sh, training=training)
    return sdt, vols, out_sh


def sp_input_of(batch_dev, out_sh):
    return {"bounds": batch_dev["bounds"], "R": batch_dev["R"], "Th": batch_dev["Th"],
            "latent_index": batch_dev["latent_index"], "out_sh": [int(s) for s in out_sh], "batch_size": 1}


def assert_close(a, b, tol, name="", rel=True):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), name + ": NaN pattern differs (%d vs %d)" % (na.sum(), nb.sum())
    err = np.abs(a[~na] - b[~na])
    if rel:
        err = err / np.maximum(1.0, np.abs(b[~na]))
    m = float(err.max(initial=0.0))
    assert m <= tol, "%s: max err %.3e > %.1e" % (name, m, tol)
    return m


# arithmetics whose summation order depends on WHICH rays share a workgroup (the fc_0-folded march contracts over the
# workgroup's voxel list): regrouping the rays moves results by rounding, not bit for bit
GROUP_DEPENDENT = ("f16f6", "auto")


def same_result(a, b, precision, tol=2e-6):
    """same_bits for the arithmetics that are invariant to how rays are grouped; equal up to rounding (`tol`, relative to
    max(1, |b|), same NaN pattern) for the others."""
    if precision not in GROUP_DEPENDENT:
        return same_bits(a, b)
    if a.shape != b.shape:
        return False
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    na, nb = torch.isnan(a), torch.isnan(b)
    if not torch.equal(na, nb):
        return False
    err = ((a - b).abs() / b.abs().clamp_min(1.0))[~na]
    return bool(err.numel() == 0 or float(err.max()) <= tol)


def same_bits(a, b):
    """Bit-exact equality that treats NaNs at the same positions as equal."""
    return a.shape == b.shape and bool(torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32)))


def load_plugin(fname, cfg=None):
    """Import neuralbody_amd/plugins/<fname> the way the reference's imp.load_source factories do, with a stand-in for
    the reference's global `lib.config.cfg` (the GPU box has no reference tree)."""
    import importlib.util
    import sys
    import types

    cfgmod = types.ModuleType("lib.config")
    cfgmod.cfg = cfg or types.SimpleNamespace(N_samples=64, perturb=1.0, raw_noise_std=0.0, white_bkgd=False, H=512, W=512, ratio=1.0)
    saved = {k: sys.modules.get(k) for k in ("lib", "lib.config")}
    sys.modules["lib"] = types.ModuleType("lib")
    sys.modules["lib.config"] = cfgmod
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("nb_plugin_" + fname.replace(".py", ""),
                                                      os.path.join(root, "neuralbody_amd", "plugins", fname))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def training_batch(seed=0, n_rand=1024, size=64, latent_index=2, dev="cuda:0"):
    """(state_dict_np, device batch) of one synthetic training iteration: n_rand random rays of one frame + rgb targets."""
    from tests import synthetic as syn

    sd = syn.make_weights(seed, num_train_frame=5)
    body = syn.make_body(seed=seed, box=(0.3, 0.5, 0.2))
    K, R, T = syn.make_camera(body, size, size, focal_factor=2.5, distance=1.5)
    ro, rd, near, far, mask = syn.host_image_rays(size, size, K, R, T, body["can_bounds"])
    rs = np.random.RandomState(seed)
    pick = rs.choice(ro.shape[0], n_rand, replace=False)  # N_rand random rays (latent_xyzc_313.yaml:66)
    batch = syn.make_batch(body, ro[pick], rd[pick], near[pick], far[pick], np.ones(n_rand, bool), latent_index=latent_index)
    batch["rgb"] = rs.uniform(0, 1, (1, n_rand, 3)).astype(np.float32)
    return sd, device_batch(batch, dev)
