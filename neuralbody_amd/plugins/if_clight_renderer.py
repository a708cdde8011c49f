"""renderer_path plugin: `Renderer(net)` bound to the reference's global cfg
(lib/networks/renderer/if_clight_renderer.py:7-9; cfg.N_samples :13, cfg.perturb :16,
cfg.raw_noise_std / cfg.white_bkgd :82)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from lib.config import cfg  # noqa: E402

from neuralbody_amd.renderer import Renderer as _Renderer  # noqa: E402


class _LiveCfg:
    """Reads the reference cfg at call time (run.py mutates cfg.perturb after import, run.py:50,82)."""

    N_samples = property(lambda self: int(cfg.N_samples))
    perturb = property(lambda self: float(cfg.perturb))
    raw_noise_std = property(lambda self: float(cfg.raw_noise_std))
    white_bkgd = property(lambda self: bool(cfg.white_bkgd))
    H = property(lambda self: int(cfg.H * cfg.ratio))  # image_rays geometry, lib/utils/render_utils.py:121-122
    W = property(lambda self: int(cfg.W * cfg.ratio))
    mesh_th = property(lambda self: float(cfg.mesh_th))


class Renderer(_Renderer):
    def __init__(self, net):
        super().__init__(net, _LiveCfg())
