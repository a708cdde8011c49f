"""The algebra the f16f6 march kernel rests on (CPU, float64): feature_fc, latent_fc and view_fc have no activation
between them (lib/networks/latent_xyzc.py:105-119), so they are ONE linear layer of fc_2's output and the encodings,
    view_w[:, :256] . latent_w[:, :256] . feature_w                      (128 x 256, over net)
    view_w[:, 256:]                                                      (128 x 90, over [viewdir PE | xyz PE])
    view_b + view_w[:, :256] . (latent_w[:, :256] . feature_b + latent_w[:, 256:] . latent_code + latent_b)
which is what nb_pack_fold_kernel (phase 2) and nb_latent_bias_kernel (second block) build.

The other fold of the kernel: trilinear interpolation (F.grid_sample, latent_xyzc.py:62-72) and fc_0 (:99) are both linear with
nothing between them, so fc_0 . interp(V) = interp(fc_0 . V): nb_fold_build pre-multiplies every active voxel with fc_0 and the
march interpolates the 256-channel result (as a matrix product with the sparse trilinear-weight matrix of its voxel list)."""
import numpy as np
import torch

from tests import synthetic as syn
from oracle import neuralbody_oracle as orc


def test_folded_colour_head_equals_the_three_layers():
    sd = orc.tensor_state_dict(syn.make_weights(3, num_train_frame=7))
    w = {k: (v[..., 0] if v.dim() == 3 else v).double() for k, v in sd.items()}
    rs = np.random.RandomState(0)
    net = torch.from_numpy(np.maximum(rs.randn(256, 50), 0.0))          # relu(fc_2(...)) of 50 samples
    pe = torch.from_numpy(rs.uniform(-1, 1, (90, 50)))                    # [view PE 27 | xyz PE 63]
    code = w["latent.weight"][4]
    # the reference's sequence (latent_xyzc.py:105-119)
    features = w["feature_fc.weight"] @ net + w["feature_fc.bias"][:, None]
    features = torch.cat([features, code[:, None].expand(128, 50)], 0)
    features = w["latent_fc.weight"] @ features + w["latent_fc.bias"][:, None]
    ref = w["view_fc.weight"] @ torch.cat([features, pe], 0) + w["view_fc.bias"][:, None]
    # the folded layer
    Vg, Vpe = w["view_fc.weight"][:, :256], w["view_fc.weight"][:, 256:]
    Lf, Ll = w["latent_fc.weight"][:, :256], w["latent_fc.weight"][:, 256:]
    W3 = Vg @ Lf @ w["feature_fc.weight"]
    b3 = w["view_fc.bias"] + Vg @ (Lf @ w["feature_fc.bias"] + Ll @ code + w["latent_fc.bias"])
    got = W3 @ net + Vpe @ pe + b3[:, None]
    assert W3.shape == (128, 256) and Vpe.shape == (128, 90)
    assert float((got - ref).abs().max()) <= 1e-12 * float(ref.abs().max())


def test_fc0_commutes_with_the_trilinear_lookup():
    """fc_0(cat_l grid_sample(V_l)) == sum_l grid_sample(fc_0[:, level l] . V_l) + b0, including zero padding (points outside
    the volume, inactive voxels) — through the reference's own grid_sample, float64."""
    rs = np.random.RandomState(1)
    sd = orc.tensor_state_dict(syn.make_weights(5, num_train_frame=3))
    w0, b0 = sd["fc_0.weight"][..., 0].double(), sd["fc_0.bias"].double()
    shapes = [(6, 10, 8), (3, 5, 4), (2, 3, 2), (1, 2, 1)]
    vols = []
    for c, sh in zip((32, 64, 128, 128), shapes):
        v = torch.from_numpy(rs.randn(1, c, *sh))
        v = v * (torch.from_numpy(rs.uniform(0, 1, (1, 1) + sh)) > 0.6)  # sparse: most voxels are exactly zero
        vols.append(v)
    g = torch.from_numpy(rs.uniform(-1.3, 1.3, (1, 1, 1, 400, 3)))  # some points outside [-1, 1]: zero padding
    feats = orc.interpolate_features(g, vols)  # [1, 352, N]
    ref = torch.nn.functional.conv1d(feats, w0[..., None], b0)
    got = b0[None, :, None].expand(1, 256, 400).clone()
    base = 0
    for v in vols:
        c = v.shape[1]
        u = torch.einsum("fc,bcdhw->bfdhw", w0[:, base:base + c], v)  # U_l = fc_0[:, level l] . V_l per voxel
        got = got + torch.nn.functional.grid_sample(u, g, padding_mode="zeros", align_corners=True).view(1, 256, 400)
        base += c
    assert base == 352
    assert float((got - ref).abs().max()) <= 1e-12 * float(ref.abs().max())


def test_trilinear_lookup_is_a_product_with_tent_weights():
    """The form the kernel evaluates: grid_sample(U)(p) = sum_v wt(v, p) U[v] with wt = the product of the three per-axis weights
    (floor + 1 - i, i - floor) of the 8 corners — ATen's corner weights — and 0 for corners outside the volume."""
    rs = np.random.RandomState(2)
    D, H, W, C = 4, 5, 6, 7
    u = torch.from_numpy(rs.randn(1, C, D, H, W))
    g = torch.from_numpy(rs.uniform(-1.2, 1.2, (1, 1, 1, 300, 3)))
    ref = torch.nn.functional.grid_sample(u, g, padding_mode="zeros", align_corners=True).view(C, 300)
    ix = (g[0, 0, 0, :, 0] + 1) / 2 * (W - 1)
    iy = (g[0, 0, 0, :, 1] + 1) / 2 * (H - 1)
    iz = (g[0, 0, 0, :, 2] + 1) / 2 * (D - 1)
    wt = torch.zeros(D * H * W, 300, dtype=torch.float64)  # the sparse matrix Wt [voxel, sample]
    x0, y0, z0 = torch.floor(ix), torch.floor(iy), torch.floor(iz)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                xx, yy, zz = x0 + dx, y0 + dy, z0 + dz
                inb = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H) & (zz >= 0) & (zz < D)
                wgt = ((x0 + 1 - ix) if dx == 0 else (ix - x0)) * ((y0 + 1 - iy) if dy == 0 else (iy - y0)) * ((z0 + 1 - iz) if dz == 0 else (iz - z0))
                lin = ((zz.clamp(0, D - 1) * H + yy.clamp(0, H - 1)) * W + xx.clamp(0, W - 1)).long()
                wt[lin[inb], torch.nonzero(inb).reshape(-1)] += wgt[inb]
    got = u.reshape(C, -1) @ wt
    assert float((got - ref).abs().max()) <= 1e-12
