"""TEST INFRASTRUCTURE ONLY — CPU restatement (plain torch/numpy) of the Neural
Body hot path, function by function, each citing the reference lines it follows
(paths are into /root/reference).  It exists so that parity can be checked on the
GPU box, where /root/reference is absent.  Never imported by the product path:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.

Pinning:
  * rows a1-a5, a7-a14 are pinned against the UNMODIFIED reference executed on CPU
    (oracle/ref_harness.py) — tests/test_reference_interop.py when /root/reference
    exists, and the committed fixtures tests/golden/*.npz (made by
    tests/golden/make_golden.py from the reference itself) everywhere.
  * row a6 (the spconv encoder): **parity unpinned** against spconv v1.2.1 @ abf0acf
    (third-party, absent, unbuildable offline); it is pinned against the dense
    stand-in oracle/spconv_standin.py driven through the reference's own
    SparseConvNet layer list (lib/networks/latent_xyzc.py:166-205).

All functions take/return torch CPU tensors; ``dtype`` may be float64 to obtain a
higher-precision truth for error budgeting.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import spconv_standin  # noqa: F401  (semantics documented there)

ENCODER_BLOCKS = [  # lib/networks/latent_xyzc.py:170-182
    ("conv0", 16, 16, 2, "subm"), ("down0", 16, 32, 1, "down"), ("conv1", 32, 32, 2, "subm"),
    ("down1", 32, 64, 1, "down"), ("conv2", 64, 64, 3, "subm"), ("down2", 64, 128, 1, "down"),
    ("conv3", 128, 128, 3, "subm"), ("down3", 128, 128, 1, "down"), ("conv4", 128, 128, 3, "subm"),
]
DENSE_AFTER = ("conv1", "conv2", "conv3", "conv4")  # :188-201


def _t(x, dtype=torch.float32):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(dtype) if x.is_floating_point() else x


def tensor_state_dict(sd_np, dtype=torch.float32):
    return {k: _t(np.asarray(v), dtype) for k, v in sd_np.items()}


# ----------------------------------------------------------------------------- a1-a3
def get_rays(H, W, K, R, T):
    """lib/utils/if_nerf/if_nerf_data_utils.py:8-21 (numpy, float64 like the reference)."""
    rays_o = -np.dot(R.T, T).ravel()
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    xy1 = np.stack([i, j, np.ones_like(i)], axis=2)
    pixel_camera = np.dot(xy1, np.linalg.inv(K).T)
    pixel_world = np.dot(pixel_camera - T.ravel(), R)
    rays_d = pixel_world - rays_o[None, None]
    rays_o = np.broadcast_to(rays_o, rays_d.shape)
    return rays_o, rays_d


def get_near_far(bounds, ray_o, ray_d):
    """lib/utils/if_nerf/if_nerf_data_utils.py:54-69."""
    norm_d = np.linalg.norm(ray_d, axis=-1, keepdims=True)
    viewdir = ray_d / norm_d
    viewdir[(viewdir < 1e-5) & (viewdir > -1e-10)] = 1e-5
    viewdir[(viewdir > -1e-5) & (viewdir < 1e-10)] = -1e-5
    tmin = (bounds[:1] - ray_o[:1]) / viewdir
    tmax = (bounds[1:2] - ray_o[:1]) / viewdir
    t1 = np.minimum(tmin, tmax)
    t2 = np.maximum(tmin, tmax)
    near = np.max(t1, axis=-1)
    far = np.min(t2, axis=-1)
    mask_at_box = near < far
    near = near[mask_at_box] / norm_d[mask_at_box, 0]
    far = far[mask_at_box] / norm_d[mask_at_box, 0]
    return near, far, mask_at_box


def image_rays(H, W, K, R, T, bounds):
    """lib/utils/render_utils.py:120-137 (H, W already multiplied by cfg.ratio)."""
    ray_o, ray_d = get_rays(H, W, K, R, T)
    ray_o = ray_o.reshape(-1, 3).astype(np.float32)
    ray_d = ray_d.reshape(-1, 3).astype(np.float32)
    near, far, mask_at_box = get_near_far(bounds, ray_o, ray_d)
    return ray_o[mask_at_box], ray_d[mask_at_box], near.astype(np.float32), far.astype(np.float32), mask_at_box


# ----------------------------------------------------------------------------- a4
def get_sampling_points(ray_o, ray_d, near, far, n_samples, t_rand=None):
    """lib/networks/renderer/if_clight_renderer.py:11-27.  ``t_rand`` ([B,P,S] in [0,1))
    replaces the reference's torch.rand when perturb>0 and the net is training."""
    t_vals = torch.linspace(0.0, 1.0, steps=n_samples).to(near)
    z_vals = near[..., None] * (1.0 - t_vals) + far[..., None] * t_vals
    if t_rand is not None:
        mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
        upper = torch.cat([mids, z_vals[..., -1:]], -1)
        lower = torch.cat([z_vals[..., :1], mids], -1)
        z_vals = lower + (upper - lower) * t_rand.to(z_vals)
    pts = ray_o[:, :, None] + ray_d[:, :, None] * z_vals[..., None]
    return pts, z_vals


# ----------------------------------------------------------------------------- a5-a6
def encode_sparse_voxels(sd, coord, out_sh, training=True, update_stats=None, relu_masks=None):
    """lib/networks/latent_xyzc.py:30-39 + SparseConvNet.forward :184-205 via the dense
    stand-in semantics (oracle/spconv_standin.py).  ``coord`` [B,N,3] int (d,h,w);
    returns the 4 NCDHW volumes.  BatchNorm1d(eps=1e-3, momentum=0.01) over ACTIVE rows
    (:215); in training mode batch statistics are used (run.py:57,89 renders in train()).
    ``update_stats``: optional dict receiving the updated running stats.
    ``relu_masks`` (gradient tests only): one bool [n_active_rows, C] per conv+BN+ReLU layer, rows in linear-voxel
    order; the ReLU then keeps exactly those entries (`x * mask`), so that a float64 reference differentiates the SAME
    piecewise-linear function as an fp32 implementation whose near-zero activations round to the other side."""
    layer_no = 0
    dtype = sd["c.weight"].dtype
    B, N = coord.shape[0], coord.shape[1]
    # Renderer.prepare_sp_input (if_clight_renderer.py:33-41): prepend the batch index
    idx = torch.arange(B).repeat_interleave(N)[:, None]
    indices = torch.cat([idx, coord.reshape(-1, 3).long()], 1)
    code = sd["c.weight"][torch.arange(0, 6890)].repeat(B, 1) if B > 1 else sd["c.weight"][torch.arange(0, 6890)]
    x = spconv_standin.SparseConvTensor(code, indices, [int(s) for s in out_sh], B)
    grid, mask = x.grid, x.mask
    vols = []
    for name, cin, cout, n, kind in ENCODER_BLOCKS:
        for j in range(n):
            w = sd["xyzc_net.%s.%d.weight" % (name, 3 * j)].permute(4, 3, 0, 1, 2)
            bnk = "xyzc_net.%s.%d" % (name, 3 * j + 1)
            if kind == "subm":
                grid = F.conv3d(grid, w, padding=1) * mask.to(dtype)
            else:
                mask = F.max_pool3d(mask.to(dtype), 3, 2, 1) > 0
                grid = F.conv3d(grid, w, stride=2, padding=1) * mask.to(dtype)
            m = mask[:, 0]
            rows = grid.permute(0, 2, 3, 4, 1)[m]
            if training:
                mean = rows.mean(0)
                var = rows.var(0, unbiased=False)
                if update_stats is not None:
                    n_rows = rows.shape[0]
                    mom = 0.01
                    update_stats[bnk + ".running_mean"] = (1 - mom) * sd[bnk + ".running_mean"] + mom * mean
                    update_stats[bnk + ".running_var"] = (1 - mom) * sd[bnk + ".running_var"] + \
                        mom * var * n_rows / max(n_rows - 1, 1)
            else:
                mean, var = sd[bnk + ".running_mean"], sd[bnk + ".running_var"]
            rows = (rows - mean) / torch.sqrt(var + 1e-3) * sd[bnk + ".weight"] + sd[bnk + ".bias"]
            rows = torch.relu(rows) if relu_masks is None else rows * relu_masks[layer_no].to(rows)
            layer_no += 1
            g = torch.zeros(grid.shape[0], *grid.shape[2:], rows.shape[1], dtype=dtype)
            g[m] = rows
            grid = g.permute(0, 4, 1, 2, 3).contiguous()
        if name in DENSE_AFTER:
            vols.append(grid)
    return vols


# ----------------------------------------------------------------------------- a7-a8
def pts_to_can_pts(pts, R, Th):
    """lib/networks/latent_xyzc.py:41-47: (p - Th) @ R (row vector times R)."""
    return torch.matmul(pts - Th, R)


def get_grid_coords(pts, bounds, out_sh, voxel_size):
    """lib/networks/latent_xyzc.py:49-60."""
    dhw = pts[..., [2, 1, 0]]
    min_dhw = bounds[:, 0, [2, 1, 0]]
    dhw = dhw - min_dhw[:, None]
    dhw = dhw / torch.tensor(voxel_size).to(dhw)
    out_sh_t = torch.tensor([int(s) for s in out_sh]).to(dhw)
    dhw = dhw / out_sh_t * 2 - 1
    return dhw[..., [2, 1, 0]]


def interpolate_features(grid_coords, feature_volume):
    """lib/networks/latent_xyzc.py:62-72 (grid_coords [B,1,1,N,3])."""
    feats = [F.grid_sample(v, grid_coords, padding_mode="zeros", align_corners=True) for v in feature_volume]
    feats = torch.cat(feats, dim=1)
    return feats.view(feats.size(0), -1, feats.size(4))


# ----------------------------------------------------------------------------- a11
def embed(x, n_freqs):
    """lib/networks/embedder.py:10-36 with log-sampled 2**linspace(0, L-1, L) bands."""
    freqs = 2.0 ** torch.linspace(0.0, n_freqs - 1, steps=n_freqs)
    out = [x]
    for f in freqs:
        out.append(torch.sin(x * f.to(x)))
        out.append(torch.cos(x * f.to(x)))
    return torch.cat(out, -1)


# ----------------------------------------------------------------------------- a9-a10
def _conv1d(sd, name, x):
    return F.conv1d(x, sd[name + ".weight"], sd[name + ".bias"])


def calculate_density(sd, wpts, feature_volume, sp, voxel_size=(0.005, 0.005, 0.005)):
    """lib/networks/latent_xyzc.py:74-89 -> [B,N,1]."""
    ppts = pts_to_can_pts(wpts, sp["R"], sp["Th"])
    g = get_grid_coords(ppts, sp["bounds"], sp["out_sh"], voxel_size)[:, None, None]
    f = interpolate_features(g, feature_volume)
    net = torch.relu(_conv1d(sd, "fc_0", f))
    net = torch.relu(_conv1d(sd, "fc_1", net))
    net = torch.relu(_conv1d(sd, "fc_2", net))
    return _conv1d(sd, "alpha_fc", net).transpose(1, 2)


def calculate_density_color(sd, wpts, viewdir, feature_volume, sp, voxel_size=(0.005, 0.005, 0.005),
                            xyz_res=10, view_res=4, relu_masks=None):
    """lib/networks/latent_xyzc.py:91-126 -> raw [B,N,4] = (rgb logits, sigma).
    ``relu_masks`` (gradient tests only): dict h1 / h2 / h3 [N,256], V [N,128] of bools replacing the four ReLUs by
    `x * mask` (see encode_sparse_voxels)."""
    def act(x, key):
        return torch.relu(x) if relu_masks is None else x * relu_masks[key].T[None].to(x)

    ppts = pts_to_can_pts(wpts, sp["R"], sp["Th"])
    g = get_grid_coords(ppts, sp["bounds"], sp["out_sh"], voxel_size)[:, None, None]
    f = interpolate_features(g, feature_volume)
    net = act(_conv1d(sd, "fc_0", f), "h1")
    net = act(_conv1d(sd, "fc_1", net), "h2")
    net = act(_conv1d(sd, "fc_2", net), "h3")
    alpha = _conv1d(sd, "alpha_fc", net)
    features = _conv1d(sd, "feature_fc", net)
    latent = sd["latent.weight"][sp["latent_index"]]
    latent = latent[..., None].expand(*latent.shape, net.size(2))
    features = _conv1d(sd, "latent_fc", torch.cat((features, latent), dim=1))
    vd = embed(viewdir, view_res).transpose(1, 2)
    lp = embed(wpts, xyz_res).transpose(1, 2)
    features = torch.cat((features, vd, lp), dim=1)
    net = act(_conv1d(sd, "view_fc", features), "V")
    rgb = _conv1d(sd, "rgb_fc", net)
    return torch.cat((rgb, alpha), dim=1).transpose(1, 2)


# ----------------------------------------------------------------------------- a13
def raw2outputs(raw, z_vals, rays_d, white_bkgd=False, noise=None):
    """lib/networks/renderer/nerf_net_utils.py:6-51; `noise` = randn * raw_noise_std (:31-35), None for raw_noise_std = 0."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    alpha = 1.0 - torch.exp(-F.relu(raw[..., 3] + (0.0 if noise is None else noise)) * dists)
    weights = alpha * torch.cumprod(
        torch.cat([torch.ones((alpha.shape[0], 1)).to(alpha), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / torch.sum(weights, -1))
    acc_map = torch.sum(weights, -1)
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


# ----------------------------------------------------------------------------- a12, a14
def render(sd, batch, n_samples=64, voxel_size=(0.005, 0.005, 0.005), training=True, t_rand=None,
           white_bkgd=False, chunk=2048, feature_volume=None, relu_masks=None):
    """lib/networks/renderer/if_clight_renderer.py:62-122 (chunked by 2048 rays).
    ``relu_masks`` (gradient tests only): {"encoder": [...], "mlp": {...}} as in encode_sparse_voxels / calculate_density_color."""
    dtype = sd["c.weight"].dtype
    b = {k: _t(v, dtype) if isinstance(v, (np.ndarray, torch.Tensor)) else v for k, v in batch.items()}
    out_sh = torch.max(b["out_sh"], dim=0)[0].tolist()
    sp = {"bounds": b["bounds"], "R": b["R"], "Th": b["Th"], "latent_index": b["latent_index"].long(),
          "out_sh": out_sh}
    if feature_volume is None:
        feature_volume = encode_sparse_voxels(sd, b["coord"], out_sh, training=training,
                                              relu_masks=None if relu_masks is None else relu_masks["encoder"])
    ray_o, ray_d, near, far = b["ray_o"], b["ray_d"], b["near"], b["far"]
    n_batch, n_pixel = ray_o.shape[:2]
    rets = []
    for i in range(0, n_pixel, chunk):
        ro, rd = ray_o[:, i:i + chunk], ray_d[:, i:i + chunk]
        tr = None if t_rand is None else t_rand[:, i:i + chunk]
        wpts, z_vals = get_sampling_points(ro, rd, near[:, i:i + chunk], far[:, i:i + chunk], n_samples, tr)
        viewdir = rd / torch.norm(rd, dim=2, keepdim=True)
        nb, npx, ns = wpts.shape[:3]
        w = wpts.view(nb, npx * ns, -1)
        v = viewdir[:, :, None].repeat(1, 1, ns, 1).contiguous().view(nb, npx * ns, -1)
        mm = None if relu_masks is None else {k: m[i * ns:(i + npx) * ns] for k, m in relu_masks["mlp"].items()}
        raw = calculate_density_color(sd, w, v, feature_volume, sp, voxel_size, relu_masks=mm)
        rgb, disp, acc, wts, depth = raw2outputs(raw.reshape(-1, ns, 4), z_vals.view(-1, ns), rd.reshape(-1, 3),
                                                 white_bkgd)
        rets.append({"rgb_map": rgb.view(nb, npx, -1), "disp_map": disp.view(nb, npx), "acc_map": acc.view(nb, npx),
                     "weights": wts.view(nb, npx, -1), "depth_map": depth.view(nb, npx), "raw": raw})
    return {k: torch.cat([r[k] for r in rets], dim=1) for k in rets[0]}


# ----------------------------------------------------------------------------- _mmsk / _msk renderers (§8(f) rank 2)
def inside_mmsk(pts, batch, H, W):
    """lib/networks/renderer/if_clight_renderer_mmsk.py:12-45: a sample survives if it projects inside ALL the
    (dilated) training-view masks.  pts [B,P,S,3] -> bool [1, P*S]."""
    sh = pts.shape
    pts = pts.view(sh[0], -1, sh[3])
    insides = []
    for nv in range(batch["Ks"].size(1)):
        R = batch["RT"][:, nv, :3, :3]
        T = batch["RT"][:, nv, :3, 3]
        pts_ = torch.matmul(pts, R.transpose(2, 1)) + T[:, None]
        pts_ = torch.matmul(pts_, batch["Ks"][:, nv].transpose(2, 1))
        pts2d = pts_[..., :2] / pts_[..., 2:]
        pts2d = pts2d.round().long()
        pts2d[..., 0] = torch.clamp(pts2d[..., 0], 0, W - 1)
        pts2d[..., 1] = torch.clamp(pts2d[..., 1], 0, H - 1)
        pts2d = pts2d[0]
        msk = batch["msks"][0, nv]
        insides.append(msk[pts2d[:, 1], pts2d[:, 0]][None].bool())
    inside = insides[0]
    for i in range(1, len(insides)):
        inside = inside * insides[i]
    return inside


def inside_msk(wpts, batch, H, W):
    """lib/networks/renderer/if_clight_renderer_msk.py:12-49: world -> SMPL space of the rendered pose -> world of the
    snapshot frame -> its camera -> its mask.  wpts [B,P,S,3] -> bool [1, P*S]."""
    can_pts = wpts - batch["Th"][:, None, None]
    can_pts = torch.matmul(can_pts, batch["R"])
    sh = can_pts.shape
    can_pts = can_pts.view(sh[0], -1, sh[3])
    pts = torch.matmul(can_pts, batch["R0_snap"].transpose(2, 1)) + batch["Th0_snap"][:, None]
    R = batch["RT"][..., :3]
    T = batch["RT"][..., 3]
    pts = torch.matmul(pts, R.transpose(2, 1)) + T[:, None]
    pts = torch.matmul(pts, batch["K"].transpose(2, 1))
    pts2d = pts[..., :2] / pts[..., 2:]
    pts2d = pts2d.round().long()
    pts2d[..., 0] = torch.clamp(pts2d[..., 0], 0, W - 1)
    pts2d[..., 1] = torch.clamp(pts2d[..., 1], 0, H - 1)
    pts2d = pts2d[0]
    msk = batch["msk"][0]
    return msk[pts2d[:, 1], pts2d[:, 0]][None].bool()


def render_masked(sd, batch, H, W, kind, n_samples=64, voxel_size=(0.005, 0.005, 0.005), training=True,
                  white_bkgd=False, chunk=2048, feature_volume=None):
    """if_clight_renderer_mmsk.py:47-94 (kind='mmsk') / _msk (kind='msk'): culled samples get raw = 0
    (full_raw = zeros, :55), the rest is the base renderer."""
    dtype = sd["c.weight"].dtype
    b = {k: _t(v, dtype) if isinstance(v, (np.ndarray, torch.Tensor)) else v for k, v in batch.items()}
    out_sh = torch.max(b["out_sh"], dim=0)[0].tolist()
    sp = {"bounds": b["bounds"], "R": b["R"], "Th": b["Th"].reshape(b["Th"].shape[0], 1, 3),
          "latent_index": b["latent_index"].long(), "out_sh": out_sh}
    if feature_volume is None:
        feature_volume = encode_sparse_voxels(sd, b["coord"], out_sh, training=training)
    ray_o, ray_d, near, far = b["ray_o"], b["ray_d"], b["near"], b["far"]
    n_pixel = ray_o.shape[1]
    rets = []
    for i in range(0, n_pixel, chunk):
        ro, rd = ray_o[:, i:i + chunk], ray_d[:, i:i + chunk]
        wpts, z_vals = get_sampling_points(ro, rd, near[:, i:i + chunk], far[:, i:i + chunk], n_samples)
        inside = inside_mmsk(wpts, b, H, W) if kind == "mmsk" else inside_msk(wpts, b, H, W)
        viewdir = rd / torch.norm(rd, dim=2, keepdim=True)
        nb, npx, ns = wpts.shape[:3]
        w = wpts.view(nb, npx * ns, -1)
        v = viewdir[:, :, None].repeat(1, 1, ns, 1).contiguous().view(nb, npx * ns, -1)
        full_raw = torch.zeros([nb, npx * ns, 4]).to(w)
        if inside.sum() > 0:
            raw = calculate_density_color(sd, w[inside][None], v[inside][None], feature_volume, sp, voxel_size)
            full_raw[inside] = raw[0]
        rgb, disp, acc, wts, depth = raw2outputs(full_raw.reshape(-1, ns, 4), z_vals.view(-1, ns), rd.reshape(-1, 3), white_bkgd)
        rets.append({"rgb_map": rgb.view(nb, npx, -1), "disp_map": disp.view(nb, npx), "acc_map": acc.view(nb, npx),
                     "weights": wts.view(nb, npx, -1), "depth_map": depth.view(nb, npx), "inside": inside.view(nb, npx, ns)})
    return {k: torch.cat([r[k] for r in rets], dim=1) for k in rets[0]}


def density_cube(sd, batch, voxel_size=(0.005, 0.005, 0.005), training=True, pad=10):
    """lib/networks/renderer/if_mesh_renderer.py:26-45 up to (not including) marching cubes: density of the lattice
    points flagged `inside`, scattered into a zero cube, zero-padded by 10 on every side.  -> float64 ndarray."""
    pts = _t(batch["pts"])
    sh = pts.shape
    inside = torch.as_tensor(batch["inside"][0]).bool()
    wpts = pts[0][inside][None]
    out_sh = np.asarray(batch["out_sh"]).max(0).tolist()
    vols = encode_sparse_voxels(sd, torch.as_tensor(batch["coord"]), out_sh, training=training)
    sp = {"R": _t(batch["R"]), "Th": _t(batch["Th"]), "bounds": _t(batch["bounds"]), "out_sh": out_sh}
    alpha = calculate_density(sd, wpts, vols, sp, voxel_size)[0, :, 0].numpy()
    cube = np.zeros(tuple(sh[1:-1]))
    cube[inside.numpy()] = alpha
    return np.pad(cube, pad, mode="constant")
