"""Synthetic Neural Body scenes: geometry, cameras and weights.

Everything here is numpy ``RandomState`` driven so that the SAME bytes are
regenerated from a seed in this container (where the golden fixtures are made
by running the reference) and on the GPU box (where only the seed travels).

Reference formulae restated here (host-side data preparation, not the hot path):
  * voxelisation of SMPL vertices -> ``coord``/``out_sh``/``bounds``:
    lib/datasets/light_stage/multi_view_dataset.py:68-118
  * batch-dict contract consumed by ``Renderer.render``: SURVEY.md §3.5
  * parameter names / shapes: lib/networks/latent_xyzc.py:9-28,166-274
"""
import math

import numpy as np

N_VERTS = 6890
CODE_DIM = 16

# (name, cin, cout, n_convs, kind) in state-dict order, lib/networks/latent_xyzc.py:170-182
ENCODER_BLOCKS = [
    ("conv0", 16, 16, 2, "subm"),
    ("down0", 16, 32, 1, "down"),
    ("conv1", 32, 32, 2, "subm"),
    ("down1", 32, 64, 1, "down"),
    ("conv2", 64, 64, 3, "subm"),
    ("down2", 64, 128, 1, "down"),
    ("conv3", 128, 128, 3, "subm"),
    ("down3", 128, 128, 1, "down"),
    ("conv4", 128, 128, 3, "subm"),
]

# name -> (out, in) of the Conv1d(k=1) MLP layers, lib/networks/latent_xyzc.py:20-28
MLP_LAYERS = {
    "fc_0": (256, 352),
    "fc_1": (256, 256),
    "fc_2": (256, 256),
    "alpha_fc": (1, 256),
    "feature_fc": (256, 256),
    "latent_fc": (256, 384),
    "view_fc": (128, 346),
    "rgb_fc": (3, 128),
}


def encoder_layer_names():
    """Yield (conv_key, bn_key, cin, cout, kind) for the 17 conv+BN+ReLU layers."""
    for name, cin, cout, n, kind in ENCODER_BLOCKS:
        c_in = cin
        for j in range(n):
            yield ("xyzc_net.%s.%d" % (name, 3 * j), "xyzc_net.%s.%d" % (name, 3 * j + 1), c_in, cout, kind)
            c_in = cout


def make_weights(seed=0, num_train_frame=230, trunk_gain=2.5, alpha_scale=20.0, alpha_bias=-3.0, rgb_scale=8.0,
                 random_bn=True):
    """A state_dict (numpy, float32) with exactly the reference's 120 keys.

    MLP layers follow nn.Conv1d's default U(-1/sqrt(fan_in), 1/sqrt(fan_in)),
    embeddings N(0,1), sparse convs N(0, 1/sqrt(27 Cin)) in spconv's
    [kD,kH,kW,Cin,Cout] layout.  The trunk / ``alpha_fc`` / ``rgb_fc``
    are rescaled so densities and colours are non-degenerate (SURVEY.md §7 'hard
    parts'): sigma spans roughly [-5, 25] with ~75 % positive, acc_map spans [0, 1].
    """
    rs = np.random.RandomState(seed)
    sd = {}
    sd["c.weight"] = rs.standard_normal((N_VERTS, CODE_DIM)).astype(np.float32)
    for conv_key, bn_key, cin, cout, _kind in encoder_layer_names():
        std = 1.0 / math.sqrt(27 * cin)
        sd[conv_key + ".weight"] = (rs.standard_normal((3, 3, 3, cin, cout)) * std).astype(np.float32)
        if random_bn:
            sd[bn_key + ".weight"] = rs.uniform(0.5, 1.5, cout).astype(np.float32)
            sd[bn_key + ".bias"] = (rs.standard_normal(cout) * 0.1).astype(np.float32)
        else:
            sd[bn_key + ".weight"] = np.ones(cout, np.float32)
            sd[bn_key + ".bias"] = np.zeros(cout, np.float32)
        if random_bn:
            sd[bn_key + ".running_mean"] = (rs.standard_normal(cout) * 0.1).astype(np.float32)
            sd[bn_key + ".running_var"] = rs.uniform(0.5, 1.5, cout).astype(np.float32)
        else:
            sd[bn_key + ".running_mean"] = np.zeros(cout, np.float32)
            sd[bn_key + ".running_var"] = np.ones(cout, np.float32)
        sd[bn_key + ".num_batches_tracked"] = np.zeros((), np.int64)
    sd["latent.weight"] = rs.standard_normal((num_train_frame, 128)).astype(np.float32)
    for name, (cout, cin) in MLP_LAYERS.items():
        bound = 1.0 / math.sqrt(cin)
        sd[name + ".weight"] = rs.uniform(-bound, bound, (cout, cin, 1)).astype(np.float32)
        sd[name + ".bias"] = rs.uniform(-bound, bound, cout).astype(np.float32)
    for name in ("fc_0", "fc_1", "fc_2"):
        sd[name + ".weight"] = sd[name + ".weight"] * np.float32(trunk_gain)
    sd["alpha_fc.weight"] = sd["alpha_fc.weight"] * np.float32(alpha_scale)
    sd["alpha_fc.bias"] = np.full(1, alpha_bias, np.float32)
    sd["rgb_fc.weight"] = sd["rgb_fc.weight"] * np.float32(rgb_scale)  # spread the colour logits
    return sd


def _rodrigues(rvec):
    theta = float(np.linalg.norm(rvec))
    if theta < 1e-12:
        return np.eye(3)
    k = np.asarray(rvec, np.float64) / theta
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(theta) * K + (1 - math.cos(theta)) * (K @ K)


def make_body(seed=0, box=(0.9, 1.7, 0.35), rh=(0.0, 0.0, 0.0), th=(0.0, 0.0, 0.0),
              voxel_size=(0.005, 0.005, 0.005), layout="uniform", n_verts=N_VERTS):
    """Synthetic SMPL frame -> the dataset-side dict of the batch contract.

    Vertices are drawn in SMPL space inside ``box`` (x,y,z extents, metres),
    placed in the world with ``world = smpl @ R^T + Th`` so that the
    reference's ``(world - Th) @ R`` recovers them
    (lib/datasets/light_stage/multi_view_dataset.py:86-91).
    """
    rs = np.random.RandomState(seed + 1000)
    box = np.asarray(box, np.float64)
    if layout == "uniform":
        smpl = (rs.uniform(0, 1, (n_verts, 3)) - 0.5) * box
    elif layout == "capsules":
        # 5-capsule body proxy: torso, 2 arms, 2 legs (surface points), occupancy ~2-3 % at level 1
        segs = [((0, -0.1, 0), (0, 0.55, 0), 0.14), ((-0.16, 0.45, 0), (-0.42, 0.0, 0), 0.05),
                ((0.16, 0.45, 0), (0.42, 0.0, 0), 0.05), ((-0.09, -0.1, 0), (-0.12, -0.82, 0), 0.07),
                ((0.09, -0.1, 0), (0.12, -0.82, 0), 0.07)]
        pts = []
        per = n_verts // len(segs)
        for si, (a, b, r) in enumerate(segs):
            n = per if si < len(segs) - 1 else n_verts - per * (len(segs) - 1)
            a = np.asarray(a, np.float64)
            b = np.asarray(b, np.float64)
            t = rs.uniform(0, 1, (n, 1))
            axis = (b - a) / np.linalg.norm(b - a)
            u = np.cross(axis, [0, 0, 1.0])
            u /= np.linalg.norm(u)
            v = np.cross(axis, u)
            ang = rs.uniform(0, 2 * math.pi, (n, 1))
            pts.append(a + t * (b - a) + r * (np.cos(ang) * u + np.sin(ang) * v))
        smpl = np.concatenate(pts, 0)
        smpl *= box / np.array([0.9, 1.7, 0.35]) * np.array([1.0, 1.0, 1.0])
    else:
        raise ValueError(layout)
    R = _rodrigues(np.asarray(rh, np.float64)).astype(np.float32)
    Th = np.asarray(th, np.float32).reshape(1, 3)
    world = (smpl.astype(np.float32) @ R.T + Th).astype(np.float32)

    # world-space AABB used for ray/box intersection (can_bounds), :75-84
    min_xyz = world.min(0).copy()
    max_xyz = world.max(0).copy()
    min_xyz[2] -= 0.05
    max_xyz[2] += 0.05
    can_bounds = np.stack([min_xyz, max_xyz], 0).astype(np.float32)

    # SMPL-space bounds + voxel coords, :86-116
    xyz = np.dot(world - Th, R).astype(np.float32)
    min_xyz = xyz.min(0).copy()
    max_xyz = xyz.max(0).copy()
    min_xyz[2] -= 0.05
    max_xyz[2] += 0.05
    bounds = np.stack([min_xyz, max_xyz], 0).astype(np.float32)
    dhw = xyz[:, [2, 1, 0]]
    min_dhw = min_xyz[[2, 1, 0]]
    max_dhw = max_xyz[[2, 1, 0]]
    vs = np.array(voxel_size)
    coord = np.round((dhw - min_dhw) / vs).astype(np.int32)
    out_sh = np.ceil((max_dhw - min_dhw) / vs).astype(np.int32)
    out_sh = (out_sh | 31) + 1
    return {
        "coord": coord, "out_sh": out_sh.astype(np.int32), "can_bounds": can_bounds, "bounds": bounds,
        "R": R, "Th": Th, "world_verts": world,
    }


def make_camera(body, H, W, focal_factor=1.2, distance=2.5, yaw=0.35, pitch=0.1):
    """Pinhole camera looking at the centre of the body's world AABB.

    Returns K (3x3 float64, like the datasets' K), R (3x3 world->camera) and
    T (3,1) with ``x_cam = R x_world + T`` (lib/utils/if_nerf/if_nerf_data_utils.py:8-21).
    """
    cb = body["can_bounds"].astype(np.float64)
    center = 0.5 * (cb[0] + cb[1])
    # camera position on a sphere around the centre
    dirv = np.array([math.sin(yaw) * math.cos(pitch), math.sin(pitch), -math.cos(yaw) * math.cos(pitch)])
    cam_pos = center + distance * dirv
    fwd = center - cam_pos
    fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, -1.0, 0.0])  # image y points down
    right = np.cross(up, fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], 0)  # rows = camera axes in world
    T = (-R @ cam_pos).reshape(3, 1)
    f = focal_factor * H
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]], np.float64)
    return K, R, T


def host_image_rays(H, W, K, R, T, bounds):
    """Host (numpy, float64->float32) full-image rays; a plain restatement used ONLY to
    build synthetic *inputs* (the device path is neuralbody_amd.raygen; the checker is
    oracle.neuralbody_oracle).  lib/utils/render_utils.py:120-137."""
    rays_o = -np.dot(R.T, T).ravel()
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    xy1 = np.stack([i, j, np.ones_like(i)], axis=2)
    pixel_camera = np.dot(xy1, np.linalg.inv(K).T)
    pixel_world = np.dot(pixel_camera - T.ravel(), R)
    rays_d = pixel_world - rays_o[None, None]
    rays_o = np.broadcast_to(rays_o, rays_d.shape)
    ray_o = rays_o.reshape(-1, 3).astype(np.float32)
    ray_d = rays_d.reshape(-1, 3).astype(np.float32)
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
    mask = near < far
    near = (near[mask] / norm_d[mask, 0]).astype(np.float32)
    far = (far[mask] / norm_d[mask, 0]).astype(np.float32)
    return ray_o[mask], ray_d[mask], near, far, mask


def make_batch(body, ray_o, ray_d, near, far, mask_at_box, latent_index=0):
    """numpy batch dict with the leading batch dim of 1."""
    return {
        "ray_o": ray_o[None].astype(np.float32), "ray_d": ray_d[None].astype(np.float32),
        "near": near[None].astype(np.float32), "far": far[None].astype(np.float32),
        "mask_at_box": mask_at_box[None],
        "coord": body["coord"][None].astype(np.int32), "out_sh": body["out_sh"][None].astype(np.int32),
        "bounds": body["bounds"][None].astype(np.float32), "R": body["R"][None].astype(np.float32),
        "Th": body["Th"][None].astype(np.float32), "latent_index": np.array([latent_index], np.int64),
    }


def full_coverage_camera(body, H, W, distance=2.5, yaw=0.35, pitch=0.1):
    """Smallest focal (in steps) for which every pixel's ray hits the AABB (throughput runs)."""
    for ff in (1.5, 2.0, 2.4, 2.8, 3.2, 3.6, 4.0, 5.0, 6.0, 8.0, 12.0):
        K, R, T = make_camera(body, H, W, focal_factor=ff, distance=distance, yaw=yaw, pitch=pitch)
        *_, mask = host_image_rays(H, W, K, R, T, body["can_bounds"])
        if mask.all():
            return K, R, T
    raise RuntimeError("no full-coverage camera found")


def make_view_masks(body, H, W, n_views=4, focal_factor=1.6, distance=2.0, dilate=3):
    """Synthetic training-view silhouettes for the mask-culled renderers (lib/networks/renderer/
    if_clight_renderer_mmsk.py:12-45): `n_views` cameras around the body, each mask = the projected vertices dilated by
    `dilate` pixels (the datasets dilate the CIHP masks by a 5x5 kernel, multi_view_demo_dataset.py:122-125).
    Returns (msks [nv,H,W] uint8, Ks [nv,3,3] f32, RT [nv,3,4] f32)."""
    from scipy import ndimage

    msks, Ks, RTs = [], [], []
    for v in range(n_views):
        K, R, T = make_camera(body, H, W, focal_factor=focal_factor, distance=distance, yaw=0.4 + 2 * math.pi * v / n_views,
                              pitch=0.05 * (v % 2))
        cam = body["world_verts"].astype(np.float64) @ R.T + T.ravel()
        uv = cam @ K.T
        uv = uv[:, :2] / uv[:, 2:]
        px = np.round(uv).astype(np.int64)
        ok = (px[:, 0] >= 0) & (px[:, 0] < W) & (px[:, 1] >= 0) & (px[:, 1] < H)
        m = np.zeros((H, W), bool)
        m[px[ok, 1], px[ok, 0]] = True
        m = ndimage.binary_dilation(m, structure=np.ones((2 * dilate + 1, 2 * dilate + 1), bool))
        msks.append(m.astype(np.uint8))
        Ks.append(K.astype(np.float32))
        RTs.append(np.concatenate([R, T], 1).astype(np.float32))
    return np.stack(msks), np.stack(Ks), np.stack(RTs)


def make_density_lattice(body, step=0.02, n_views=3, H=64, W=64, dilate=2):
    """The mesh-extraction query set of lib/datasets/light_stage/multi_view_mesh_dataset.py:142-158: a regular lattice
    over the world bounding box (np.arange per axis, 'ij' meshgrid) and the `inside` bitmap = lattice points that project
    inside every synthetic silhouette (the dataset's prepare_inside_pts, :100-140, done here in numpy).
    Returns (pts [X,Y,Z,3] f32, inside [X,Y,Z] uint8)."""
    cb = body["can_bounds"].astype(np.float32)
    axes = [np.arange(cb[0, a], cb[1, a] + step, step) for a in range(3)]
    pts = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).astype(np.float32)
    msks, Ks, RT = make_view_masks(body, H, W, n_views=n_views, focal_factor=1.8, distance=1.6, dilate=dilate)
    flat = pts.reshape(-1, 3).astype(np.float64)
    inside = np.ones(flat.shape[0], bool)
    for v in range(n_views):
        cam = flat @ RT[v, :, :3].astype(np.float64).T + RT[v, :, 3].astype(np.float64)
        uv = cam @ Ks[v].astype(np.float64).T
        px = np.round(uv[:, :2] / uv[:, 2:]).astype(np.int64)
        x, y = np.clip(px[:, 0], 0, W - 1), np.clip(px[:, 1], 0, H - 1)
        inside &= msks[v][y, x] != 0
    return pts, inside.reshape(pts.shape[:-1]).astype(np.uint8)
