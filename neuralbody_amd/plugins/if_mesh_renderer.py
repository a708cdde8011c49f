"""renderer_path plugin for mesh extraction (lib/networks/renderer/if_mesh_renderer.py): `Renderer(net)` bound to the
reference's global cfg; selected by the `mesh_cfg` overlay of the shipped configs
(e.g. configs/zju_mocap_exp/latent_xyzc_313.yaml:140-148)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from neuralbody_amd.plugins.if_clight_renderer import _LiveCfg  # noqa: E402
from neuralbody_amd.renderer import RendererMesh as _Renderer  # noqa: E402


class Renderer(_Renderer):
    def __init__(self, net):
        super().__init__(net, _LiveCfg())
