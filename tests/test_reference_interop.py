"""Interop with the reference's own code (CPU; needs /root/reference, skipped on the GPU box):
  * a checkpoint written by the reference's save_model from the reference Network loads into our Network
    through the reference's load_network (lib/utils/net_utils.py:326-380), key for key, value for value;
  * the plugin files resolve through the reference's make_network / make_renderer factories
    (lib/networks/make_network.py:5-9, lib/networks/renderer/make_renderer.py:5-9) and build our classes."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_harness as rh

pytestmark = pytest.mark.skipif(not rh.available(), reason="reference tree not present")

PLUGINS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neuralbody_amd", "plugins")


def test_reference_checkpoint_round_trip(tmp_path):
    from tests import synthetic as syn
    from neuralbody_amd.network import Network

    ns = rh.load()
    import lib.utils.net_utils as nu  # the reference's module (importable once rh.load() ran)

    sd = syn.make_weights(3, num_train_frame=9)
    ref_net = rh.make_reference_network(sd)

    class _Dummy:  # optimizer / scheduler / recorder stand-ins with the state_dict surface save_model uses
        def state_dict(self):
            return {}

    nu.save_model(ref_net, _Dummy(), _Dummy(), _Dummy(), str(tmp_path), epoch=7, last=True)
    ours = Network(num_train_frame=9)
    next_epoch = nu.load_network(ours, str(tmp_path), strict=True)
    assert next_epoch == 8
    got = ours.state_dict()
    assert list(got) == list(ref_net.state_dict()), "state_dict key ORDER differs from the reference"
    for k, v in ref_net.state_dict().items():
        assert torch.equal(got[k], v), k
    # and back: our weights load into the reference module
    ref_net.load_state_dict(ours.state_dict(), strict=True)


def test_plugins_resolve_through_reference_factories():
    ns = rh.load()
    cfg = ns.cfg
    saved = (cfg.network_module, cfg.network_path, cfg.renderer_module, cfg.renderer_path)
    cwd = os.getcwd()
    os.chdir(ns.root)
    try:
        cfg.network_module, cfg.network_path = "lib.networks.latent_xyzc_hip", os.path.join(PLUGINS, "latent_xyzc.py")
        cfg.renderer_module = "lib.networks.renderer.if_clight_renderer_hip"
        cfg.renderer_path = os.path.join(PLUGINS, "if_clight_renderer.py")
        net = ns.make_network(cfg)
        ren = ns.make_renderer(cfg, net)
        culled = {}
        # the novel-view and mesh overlays' renderers (latent_xyzc_313.yaml:95,140-148, snapshot_f3c.yaml:88)
        for kind, fname in (("mmsk", "if_clight_renderer_mmsk"), ("msk", "if_clight_renderer_msk"), ("mesh", "if_mesh_renderer")):
            cfg.renderer_module = "lib.networks.renderer.%s_hip" % fname
            cfg.renderer_path = os.path.join(PLUGINS, fname + ".py")
            culled[kind] = ns.make_renderer(cfg, net)
    finally:
        cfg.network_module, cfg.network_path, cfg.renderer_module, cfg.renderer_path = saved
        os.chdir(cwd)
    from neuralbody_amd.network import Network
    from neuralbody_amd.renderer import Renderer, RendererMesh, RendererMmsk, RendererMsk

    assert isinstance(net, Network) and isinstance(ren, Renderer)
    assert isinstance(culled["mmsk"], RendererMmsk) and isinstance(culled["msk"], RendererMsk)
    assert isinstance(culled["mesh"], RendererMesh) and culled["mesh"].cfg.mesh_th == cfg.mesh_th
    assert net.latent.weight.shape[0] == cfg.num_train_frame
    assert ren.cfg.N_samples == cfg.N_samples and ren.cfg.H == int(cfg.H * cfg.ratio)
    for name in ("encode_sparse_voxels", "calculate_density", "calculate_density_color", "forward"):
        assert callable(getattr(net, name))
    for name in ("get_sampling_points", "prepare_sp_input", "get_density_color", "get_pixel_value", "render"):
        assert callable(getattr(ren, name))


def test_trainer_plugin_resolves_through_reference_wrapper_factory():
    """lib/train/trainers/make_trainer.py:5-9: `_wrapper_factory(cfg, network)` imp.load_source's cfg.trainer_path and builds
    NetworkWrapper(network).  (make_trainer itself goes on to Trainer(...), which moves the module to cuda:0 — not on this box.)"""
    ns = rh.load()
    cfg = ns.cfg
    import importlib
    import sys

    importlib.import_module("lib.train.trainers.make_trainer")
    mt = sys.modules["lib.train.trainers.make_trainer"]  # the package re-exports a function under the module's name

    saved = (cfg.network_module, cfg.network_path, cfg.renderer_module, cfg.renderer_path, cfg.trainer_module, cfg.trainer_path)
    cwd = os.getcwd()
    os.chdir(ns.root)
    try:
        cfg.network_module, cfg.network_path = "lib.networks.latent_xyzc_hip", os.path.join(PLUGINS, "latent_xyzc.py")
        cfg.trainer_module, cfg.trainer_path = "lib.train.trainers.if_nerf_clight_hip", os.path.join(PLUGINS, "if_nerf_clight.py")
        net = ns.make_network(cfg)
        wrapper = mt._wrapper_factory(cfg, net)
    finally:
        (cfg.network_module, cfg.network_path, cfg.renderer_module, cfg.renderer_path, cfg.trainer_module, cfg.trainer_path) = saved
        os.chdir(cwd)
    from neuralbody_amd.network import Network
    from neuralbody_amd.renderer import Renderer

    assert type(wrapper).__name__ == "NetworkWrapper" and isinstance(wrapper, torch.nn.Module)
    assert isinstance(wrapper.net, Network) and wrapper.net is net and isinstance(wrapper.renderer, Renderer)
    assert wrapper.renderer.cfg.N_samples == cfg.N_samples
    # the Trainer drives exactly this surface (trainer.py:46-52): parameters() for clip/Adam, forward(batch) -> 4-tuple
    assert len(list(wrapper.parameters())) == 69
    assert set(dict(wrapper.named_parameters())) == {"net." + k for k, _ in net.named_parameters()}


def test_the_reference_itself_cannot_run_other_embedder_resolutions():
    """Why Network(xyz_res != 10 or view_res != 4) is refused: the reference's Network hard-codes view_fc = Conv1d(346, 128, 1)
    (latent_xyzc.py:27) while its embedders take their width from cfg.xyz_res / cfg.view_res at import (embedder.py:53-54) — with
    xyz_res = 8 the decoder's concatenation has 334 channels and view_fc raises.  Run in a fresh interpreter: the embedder globals
    are frozen at the first import of the reference."""
    import subprocess
    import sys

    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from oracle import ref_harness as rh\n"
        "from tests.golden import scenes\n"
        "ns = rh.load(opts=('perturb', '0', 'xyz_res', '8'))\n"
        "assert ns.embedder.xyz_dim == 3 + 6 * 8, ns.embedder.xyz_dim\n"
        "r, sd, body, batch, cam, _ = scenes.build('small')\n"
        "net = rh.make_reference_network(sd)\n"
        "assert tuple(net.view_fc.weight.shape) == (128, 346, 1)\n"
        "ren = rh.make_reference_renderer(net)\n"
        "try:\n"
        "    with torch.no_grad():\n"
        "        ren.render(rh.torch_batch(batch))\n"
        "except RuntimeError as e:\n"
        "    print('REFERENCE RAISED:', str(e)[:120])\n"
        "else:\n"
        "    print('REFERENCE RAN')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "REFERENCE RAISED:" in out.stdout and "REFERENCE RAN" not in out.stdout, out.stdout
