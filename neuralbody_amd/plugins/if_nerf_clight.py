"""trainer_path plugin: `NetworkWrapper(net)` with the reference's forward contract
(lib/train/trainers/if_nerf_clight.py:8-37): returns (ret, loss, scalar_stats, image_stats) with the
masked MSE of rgb_map against batch['rgb'].  Under torch.enable_grad() with trainable parameters
Renderer.render takes the differentiable HIP path (neuralbody_amd/training.py), so loss.backward() fills the
gradients of every parameter; under no_grad it is the fused inference march."""
import os
import sys

import torch
import torch.nn as nn

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from neuralbody_amd.plugins.if_clight_renderer import Renderer  # noqa: E402


class NetworkWrapper(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net
        self.renderer = Renderer(self.net)
        self.img2mse = lambda x, y: torch.mean((x - y) ** 2)

    def forward(self, batch):
        ret = self.renderer.render(batch)
        scalar_stats = {}
        loss = 0
        mask = batch["mask_at_box"]
        img_loss = self.img2mse(ret["rgb_map"][mask], batch["rgb"][mask])
        scalar_stats.update({"img_loss": img_loss})
        loss += img_loss
        scalar_stats.update({"loss": loss})
        image_stats = {}
        return ret, loss, scalar_stats, image_stats
