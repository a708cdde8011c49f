"""TEST INFRASTRUCTURE ONLY — drives the UNMODIFIED reference Python
(/root/reference) on CPU so that golden vectors and oracle checks are pinned by
the reference's own code.

Only usable where /root/reference exists (the build container); nothing that
runs on the GPU box may import this.  Used by tests/golden/make_golden.py and by
the ``not gpu`` tests that cross-check oracle/neuralbody_oracle.py.

What is stubbed, and why (none of it is on the hot path):
  * ``open3d``  — imported, unused, at lib/config/config.py:1
  * ``cv2``, ``trimesh`` — top-level imports of lib/utils/if_nerf/if_nerf_data_utils.py:3,5
  * ``tensorboardX``, ``termcolor`` — logging imports of lib/train/recorder.py:3, lib/utils/net_utils.py
  * ``spconv`` — absent third-party CUDA library; replaced by oracle/spconv_standin.py
  * ``sys.argv`` — lib/config/config.py:176-187 parses argv at import time
"""
import os
import sys
import types

REF_ROOT = "/root/reference"
_state = {}


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "lib", "networks"))


def load(cfg_file="configs/snapshot_exp/snapshot_f3c.yaml", opts=("perturb", "0")):
    """Import the reference once per process; returns a namespace of its hot-path symbols."""
    if "ns" in _state:
        return _state["ns"]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    here = os.path.dirname(os.path.abspath(__file__))
    if os.path.dirname(here) not in sys.path:
        sys.path.insert(0, os.path.dirname(here))
    from oracle import spconv_standin

    for name in ("open3d", "cv2", "trimesh", "termcolor"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if "tensorboardX" not in sys.modules:  # lib/train/recorder.py:3 (logging only)
        tbx = types.ModuleType("tensorboardX")
        tbx.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None})
        sys.modules["tensorboardX"] = tbx
    sys.modules["termcolor"].colored = lambda s, *a, **k: s
    sys.modules["spconv"] = spconv_standin
    saved_argv, saved_cwd = list(sys.argv), os.getcwd()
    saved_cvd = os.environ.get("CUDA_VISIBLE_DEVICES")
    os.chdir(REF_ROOT)
    sys.path.insert(0, REF_ROOT)
    sys.argv = ["ref_harness", "--cfg_file", cfg_file] + list(opts)
    try:
        from lib.config import cfg
        from lib.networks import make_network
        from lib.networks.renderer import make_renderer
        from lib.networks.renderer import nerf_net_utils
        from lib.networks import embedder
        from lib.utils.if_nerf import if_nerf_data_utils
        from lib.utils import render_utils
        from lib.train.trainers import if_nerf_clight
    finally:
        sys.argv = saved_argv
        os.chdir(saved_cwd)
        # lib/config/config.py:137 overwrites CUDA_VISIBLE_DEVICES; undo
        if saved_cvd is None:
            os.environ.pop("CUDA_VISIBLE_DEVICES", None)
        else:
            os.environ["CUDA_VISIBLE_DEVICES"] = saved_cvd
    ns = types.SimpleNamespace(
        cfg=cfg, make_network=make_network, make_renderer=make_renderer, raw2outputs=nerf_net_utils.raw2outputs,
        embedder=embedder, get_rays=if_nerf_data_utils.get_rays, get_near_far=if_nerf_data_utils.get_near_far,
        image_rays=render_utils.image_rays, NetworkWrapper=if_nerf_clight.NetworkWrapper, root=REF_ROOT)
    _state["ns"] = ns
    return ns


def make_reference_network(state_dict_np, train_mode=True):
    """Reference ``Network`` (lib/networks/latent_xyzc.py:9) loaded with a numpy state dict."""
    import torch

    ns = load()
    cwd = os.getcwd()
    os.chdir(REF_ROOT)  # make_network loads the plugin by relative path (lib/networks/make_network.py:5-9)
    try:
        num_frames = state_dict_np["latent.weight"].shape[0]
        ns.cfg.num_train_frame = int(num_frames)
        net = ns.make_network(ns.cfg)
    finally:
        os.chdir(cwd)
    sd = {k: torch.from_numpy(v.copy()) if v.ndim else torch.tensor(v) for k, v in state_dict_np.items()}
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    net.train(train_mode)  # run.py:57,89 evaluates/visualises in train() mode
    return net


def make_reference_renderer(net):
    ns = load()
    cwd = os.getcwd()
    os.chdir(REF_ROOT)
    try:
        return ns.make_renderer(ns.cfg, net)
    finally:
        os.chdir(cwd)


def torch_batch(batch_np):
    import torch

    return {k: torch.from_numpy(v) for k, v in batch_np.items()}
