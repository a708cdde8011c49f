"""Host-side logic of Renderer that needs no GPU: the ray tile order and the out_sh read-back cache."""
import torch

from neuralbody_amd import ops
from neuralbody_amd.network import Network
from neuralbody_amd.renderer import RenderConfig, Renderer


def _renderer(H, W):
    return Renderer(Network(num_train_frame=3), RenderConfig(N_samples=8, perturb=0.0, H=H, W=W))


def test_tile_order_of_a_full_coverage_view_needs_no_mask_readback():
    """Every pixel a ray: the order is a function of the image geometry; it equals what the general path (non-zeros of the
    mask -> tile keys -> argsort) gives, is a permutation, is computed once per geometry / ray range, and 32 consecutive slots
    hold one 8x4 pixel tile."""
    H, W = 16, 24
    r = _renderer(H, W)
    n = H * W
    full = r._tile_order({"mask_at_box": torch.ones(1, n, dtype=torch.bool)}, n, 0, n)
    ref = ops.tile_order(torch.nonzero(torch.ones(n, dtype=torch.bool)).reshape(-1), W)
    assert torch.equal(full, ref) and full.dtype == torch.int32
    assert torch.equal(torch.sort(full.long()).values, torch.arange(n))
    tile = full[:32].long()
    assert int((tile // W).max() - (tile // W).min()) == 3 and int((tile % W).max() - (tile % W).min()) == 7
    again = r._tile_order({"mask_at_box": torch.ones(1, n, dtype=torch.bool)}, n, 0, n)  # another mask tensor, same geometry
    assert again is full
    part = r._tile_order({"mask_at_box": torch.ones(1, n, dtype=torch.bool)}, n, 64, 320)  # a rank's share of the rays
    assert torch.equal(torch.sort(part.long()).values, torch.arange(256)) and part is not full


def test_tile_order_of_a_partial_mask_follows_the_mask():
    H, W = 16, 24
    r = _renderer(H, W)
    mask = torch.zeros(H, W, dtype=torch.bool)
    mask[2:14, 3:20] = True
    n = int(mask.sum())
    batch = {"mask_at_box": mask.reshape(1, -1)}
    order = r._tile_order(batch, n, 0, n)
    assert torch.equal(order, ops.tile_order(torch.nonzero(mask.reshape(-1)).reshape(-1), W))
    assert r._tile_order(batch, n, 0, n) is order          # same tensor object, same version: cached
    assert r._tile_order(batch, n - 1, 0, n - 1) is None   # ray count and mask disagree: list order


def test_out_sh_is_read_once_per_tensor_version():
    r = _renderer(8, 8)
    t = torch.tensor([[96, 320, 192], [64, 352, 128]], dtype=torch.int32)
    assert r._host_out_sh(t) == [96, 352, 192]
    got = r._host_out_sh(t)
    got[0] = -1                                              # callers may edit their copy
    assert r._host_out_sh(t) == [96, 352, 192]
    t[0, 0] = 128                                            # in-place write bumps the version: read again
    assert r._host_out_sh(t) == [128, 352, 192]
    assert r._host_out_sh(t.clone()) == [128, 352, 192]      # another tensor object: read again (and cached in turn)
