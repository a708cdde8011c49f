"""network_path plugin: `Network()` with the reference's zero-argument constructor, reading the
reference's global cfg (lib/networks/latent_xyzc.py:9-16 reads cfg.num_train_frame; :54 cfg.voxel_size;
lib/networks/embedder.py:53-54 cfg.xyz_res / cfg.view_res)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from lib.config import cfg  # noqa: E402  (the reference's config module; run from the reference checkout)

from neuralbody_amd.network import Network as _Network  # noqa: E402


class Network(_Network):
    def __init__(self):
        super().__init__(num_train_frame=cfg.num_train_frame, voxel_size=cfg.voxel_size, xyz_res=cfg.xyz_res,
                         view_res=cfg.view_res)
