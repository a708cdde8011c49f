"""Shared builders for the GPU parity tests: golden scenes -> device batch / Network / oracle."""
import os

import numpy as np
import torch

from tests.golden import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RGB_TOL = 1e-4  # BASELINE.json north_star: <= 1e-4 RGB L-inf against the reference renderer
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


def same_bits(a, b):
    """Bit-exact equality that treats NaNs at the same positions as equal."""
    return a.shape == b.shape and bool(torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32)))
