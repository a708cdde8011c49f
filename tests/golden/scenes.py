"""Scene recipes shared by the golden generator (make_golden.py, runs the reference)
and by the tests (which regenerate the identical inputs from the seeds)."""
import numpy as np

from tests import synthetic as syn

_SMALL_BODY = dict(seed=0, box=(0.3, 0.5, 0.2), rh=(0.1, 0.2, -0.1), th=(0.05, -0.1, 0.2))
_SMALL_CAM = dict(H=32, W=32, focal_factor=2.5, distance=1.5)

SCENES = {
    # name: recipe.  mode 'train' == run.py's evaluate/visualise mode (net.train(), perturb=0)
    "small": dict(weights_seed=0, num_train_frame=7, body=_SMALL_BODY, cam=_SMALL_CAM, n_samples=64,
                  latent_index=3, mode="train", perturb=False, white_bkgd=False, probes=True),
    "small_s128": dict(weights_seed=0, num_train_frame=7, body=_SMALL_BODY,
                       cam=dict(H=16, W=16, focal_factor=2.5, distance=1.5), n_samples=128,
                       latent_index=0, mode="train", perturb=False, white_bkgd=False, probes=False),
    "small_perturb": dict(weights_seed=0, num_train_frame=7, body=_SMALL_BODY, cam=_SMALL_CAM, n_samples=64,
                          latent_index=6, mode="train", perturb=True, white_bkgd=False, probes=False),
    "small_eval": dict(weights_seed=0, num_train_frame=7, body=_SMALL_BODY, cam=_SMALL_CAM, n_samples=64,
                       latent_index=1, mode="eval", perturb=False, white_bkgd=True, probes=True,
                       weights_kw=dict(alpha_bias=0.7)),  # eval-mode BN squashes the features: centre sigma on 0
    "full": dict(weights_seed=1, num_train_frame=5,
                 body=dict(seed=1, box=(0.9, 1.7, 0.35), rh=(0.3, -0.2, 0.1), th=(0.1, 0.2, -0.3), layout="capsules"),
                 cam=dict(H=24, W=20, focal_factor=1.3, distance=2.5), n_samples=64,
                 latent_index=2, mode="train", perturb=False, white_bkgd=False, probes=True,
                 weights_kw=dict(alpha_bias=2.0)),  # empty space (all-zero features) gets sigma > 0: fog with holes
    # ~20 % of the samples have sigma > 0: transmittance decays over many samples (stresses the compositing)
    "small_dense": dict(weights_seed=2, num_train_frame=7, body=_SMALL_BODY, cam=_SMALL_CAM, n_samples=64,
                        latent_index=5, mode="train", perturb=False, white_bkgd=True, probes=False,
                        weights_kw=dict(alpha_bias=0.0, alpha_scale=12.0)),
}
N_PROBES = 160
RAW_RAY_STRIDE = 8


def build(name):
    """-> (recipe, state_dict_np, body, batch_np, cam(K,R,T,H,W), t_rand or None)"""
    r = SCENES[name]
    sd = syn.make_weights(r["weights_seed"], num_train_frame=r["num_train_frame"], **r.get("weights_kw", {}))
    body = syn.make_body(**r["body"])
    c = r["cam"]
    K, R, T = syn.make_camera(body, c["H"], c["W"], focal_factor=c["focal_factor"], distance=c["distance"])
    ray_o, ray_d, near, far, mask = syn.host_image_rays(c["H"], c["W"], K, R, T, body["can_bounds"])
    batch = syn.make_batch(body, ray_o, ray_d, near, far, mask, latent_index=r["latent_index"])
    t_rand = None
    if r["perturb"]:
        t_rand = np.random.RandomState(77).uniform(0, 1, (1, ray_o.shape[0], r["n_samples"])).astype(np.float32)
        t_rand = np.minimum(t_rand, np.float32(1.0 - 2 ** -24))
    return r, sd, body, batch, (K, R, T, c["H"], c["W"]), t_rand


def probe_indices(mask_flat, n=N_PROBES, seed=123):
    """Deterministic probe voxels: 3/4 from the active set, 1/4 anywhere."""
    rs = np.random.RandomState(seed)
    act = np.flatnonzero(mask_flat)
    n_act = min(len(act), (3 * n) // 4)
    a = act[rs.choice(len(act), n_act, replace=False)] if n_act else np.zeros(0, np.int64)
    b = rs.randint(0, mask_flat.size, n - n_act)
    return np.concatenate([a, b]).astype(np.int64)


# training-step recipe: N_rand random rays of one frame, stratified jitter, MSE against `rgb`
TRAIN = dict(weights_seed=4, num_train_frame=6, body=_SMALL_BODY, cam=dict(H=48, W=48, focal_factor=2.5, distance=1.5),
             n_samples=64, n_rand=256, latent_index=4, weights_kw=dict(alpha_bias=0.0, alpha_scale=12.0))
GRAD_PROBES = 6


def build_train():
    """-> (recipe, state_dict_np, batch_np incl. 'rgb' target and 'mask_at_box' over the sampled rays, t_rand)"""
    r = TRAIN
    sd = syn.make_weights(r["weights_seed"], num_train_frame=r["num_train_frame"], **r["weights_kw"])
    body = syn.make_body(**r["body"])
    c = r["cam"]
    K, R, T = syn.make_camera(body, c["H"], c["W"], focal_factor=c["focal_factor"], distance=c["distance"])
    ray_o, ray_d, near, far, mask = syn.host_image_rays(c["H"], c["W"], K, R, T, body["can_bounds"])
    rs = np.random.RandomState(31)
    pick = np.sort(rs.choice(ray_o.shape[0], r["n_rand"], replace=False))
    batch = syn.make_batch(body, ray_o[pick], ray_d[pick], near[pick], far[pick], np.ones(r["n_rand"], bool),
                           latent_index=r["latent_index"])
    batch["rgb"] = rs.uniform(0, 1, (1, r["n_rand"], 3)).astype(np.float32)
    t_rand = np.minimum(rs.uniform(0, 1, (1, r["n_rand"], r["n_samples"])).astype(np.float32), np.float32(1.0 - 2 ** -24))
    return r, sd, batch, t_rand


# "trained-like" weights: the reference's NetworkWrapper + Adam run for TRAINED["steps"] steps on the `small`
# scene against a smooth synthetic target image; the decoder MLP, the per-frame latent codes and the vertex codes are optimised
# (the parameters whose distribution the six-bit cross terms are sensitive to: 542 k floats, stored in the fixture), the sparse
# convolutions keep their seed-0 values (3.8 M floats: they would make the fixture 17 MB).
TRAINED = dict(base="small", steps=300, n_rand=256, lr=5e-4, seed=11)
TRAINED_PREFIXES = ("fc_0.", "fc_1.", "fc_2.", "alpha_fc.", "feature_fc.", "latent_fc.", "view_fc.", "rgb_fc.", "latent.", "c.")


def trained_target(H, W, mask):
    """Smooth RGB target over the image (values in [0.1, 0.9]) for the rays inside the box mask: [n_rays, 3]."""
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32) / H, np.arange(W, dtype=np.float32) / W, indexing="ij")
    img = np.stack([0.5 + 0.4 * np.sin(6.0 * xx + 1.0), 0.5 + 0.4 * np.cos(5.0 * yy - 0.5), 0.5 + 0.4 * np.sin(4.0 * (xx + yy))], -1)
    return img.reshape(-1, 3)[mask.reshape(-1)].astype(np.float32)


def build_trained(trained_params):
    """`small` with the optimised parameters of tests/golden/scene_small_trained.npz patched in -> same tuple as build()."""
    r, sd, body, batch, cam, t_rand = build(TRAINED["base"])
    sd = dict(sd)
    for k, v in trained_params.items():
        assert k in sd and sd[k].shape == v.shape, k
        sd[k] = v
    return r, sd, body, batch, cam, t_rand


def grad_probe_indices(shape, n=GRAD_PROBES, seed=5):
    size = int(np.prod(shape)) if len(shape) else 1
    return np.random.RandomState(seed + size % 1000).randint(0, size, n)


# mask-culled renderers (§8(f) rank 2): 'mmsk' = multi-view ZJU novel view, 'msk' = monocular People-Snapshot novel view
# a capsule "body" (thin limbs inside a mostly empty bounding box) so that the silhouettes really cull samples
_CAPSULE_BODY = dict(seed=5, box=(0.45, 0.85, 0.18), rh=(0.1, 0.2, -0.1), th=(0.05, -0.1, 0.2), layout="capsules")
MASKED = {
    "mmsk": dict(weights_seed=2, num_train_frame=7, body=_CAPSULE_BODY, cam=dict(H=48, W=48, focal_factor=1.8, distance=1.6),
                 n_samples=64, latent_index=1, n_views=4, weights_kw=dict(alpha_bias=1.0, alpha_scale=12.0)),
    "msk": dict(weights_seed=2, num_train_frame=7, body=_CAPSULE_BODY, cam=dict(H=48, W=48, focal_factor=1.8, distance=1.6),
                n_samples=64, latent_index=2, n_views=1, weights_kw=dict(alpha_bias=1.0, alpha_scale=12.0)),
}


def build_masked(kind):
    """-> (recipe, state_dict_np, batch_np with the extra keys of the _mmsk / _msk datasets, (H, W))"""
    r = MASKED[kind]
    sd = syn.make_weights(r["weights_seed"], num_train_frame=r["num_train_frame"], **r["weights_kw"])
    body = syn.make_body(**r["body"])
    c = r["cam"]
    K, R, T = syn.make_camera(body, c["H"], c["W"], focal_factor=c["focal_factor"], distance=c["distance"], yaw=-0.3)
    ray_o, ray_d, near, far, mask = syn.host_image_rays(c["H"], c["W"], K, R, T, body["can_bounds"])
    batch = syn.make_batch(body, ray_o, ray_d, near, far, mask, latent_index=r["latent_index"])
    msks, Ks, RT = syn.make_view_masks(body, c["H"], c["W"], n_views=r["n_views"], focal_factor=1.8, distance=1.6, dilate=1)
    if kind == "mmsk":  # multi_view_demo_dataset.py:176
        batch.update(msks=msks[None], Ks=Ks[None], RT=RT[None])
    else:  # monocular_demo_dataset.py:117-142: Th is [3]; the snapshot frame's own pose places the mask
        batch["Th"] = batch["Th"].reshape(1, 3)
        batch.update(msk=(msks[0] * 255)[None].astype(np.uint8), K=Ks[0][None], RT=RT[0][None],
                     R0_snap=body["R"][None].astype(np.float32), Th0_snap=body["Th"].reshape(1, 3).astype(np.float32))
    return r, sd, batch, (c["H"], c["W"])


# density lattice for mesh extraction (§8(f) rank 4): if_mesh_renderer.py over a coarse lattice of the capsule body
MESH = dict(weights_seed=6, num_train_frame=5, body=_CAPSULE_BODY, step=0.025, latent_index=3,
            weights_kw=dict(alpha_bias=1.0, alpha_scale=12.0))


def build_mesh():
    """-> (recipe, state_dict_np, batch_np with 'pts' [1,X,Y,Z,3] and 'inside' [1,X,Y,Z] — the keys of
    multi_view_mesh_dataset.py:160-181)"""
    r = MESH
    sd = syn.make_weights(r["weights_seed"], num_train_frame=r["num_train_frame"], **r["weights_kw"])
    body = syn.make_body(**r["body"])
    pts, inside = syn.make_density_lattice(body, step=r["step"])
    batch = {
        "pts": pts[None], "inside": inside[None],
        "coord": body["coord"][None].astype(np.int32), "out_sh": body["out_sh"][None].astype(np.int32),
        "bounds": body["bounds"][None].astype(np.float32), "R": body["R"][None].astype(np.float32),
        "Th": body["Th"][None].astype(np.float32), "latent_index": np.array([r["latent_index"]], np.int64),
        "wbounds": body["can_bounds"][None].astype(np.float32), "frame_index": np.array([0], np.int64),
    }
    return r, sd, batch


# novel-view driving (§8(f) rank 3): training cameras for gen_path / load_cam, a posed body for the rotating-SMPL frames
NOVEL = dict(body=dict(seed=8, box=(0.45, 0.85, 0.18), rh=(0.3, -0.5, 0.2), th=(0.1, 0.05, 1.9), layout="capsules"),
             n_cams=5, num_render_views=6, ratio=0.5, turntable_steps=(0, 17, 100))


def build_novel():
    """-> (recipe, body, cams dict as stored in the datasets' annots (K, R, T in millimetres), center)"""
    r = NOVEL
    body = syn.make_body(**r["body"])
    cams = {"K": [], "R": [], "T": []}
    for v in range(r["n_cams"]):
        K, R, T = syn.make_camera(body, 96, 128, focal_factor=1.4, distance=2.2 + 0.1 * v, yaw=-0.9 + 0.45 * v,
                                  pitch=0.05 * ((v % 3) - 1))
        cams["K"].append(K.tolist())
        cams["R"].append(R.tolist())
        cams["T"].append((T.reshape(3, 1) * 1000.0).tolist())
    center = body["world_verts"].mean(0).astype(np.float64)
    return r, body, cams, center


# The BENCH scene (bench.build_scene / build_poses: body seed 0, weights seed 0 with 230 frames, the full-coverage camera of pose 1 —
# the view bench.py's parity leg and tests/test_gpu_fullsize.py check) as a ray-list batch of N_BENCH_RAYS rays spread over the
# image: what the unmodified reference renders at the headline size (make_golden.run_bench), so that parity AT SIZE is held
# against the reference itself and not only against the port.
BENCH_POSE = 1
N_BENCH_RAYS = 384
BENCH = {"512x64": dict(size=512, n_samples=64), "1024x128": dict(size=1024, n_samples=128)}


def build_bench(tag):
    """-> (recipe, state_dict_np, body, batch_np of the picked rays, picked ray indices into the H*W rays of the view)"""
    r = BENCH[tag]
    H = W = r["size"]
    sd = syn.make_weights(0, num_train_frame=230)
    body = syn.make_body(seed=0)
    i = BENCH_POSE
    K, R, T = syn.full_coverage_camera(body, H, W, yaw=0.35 + 0.12 * i, pitch=0.1 - 0.03 * i)
    ray_o, ray_d, near, far, mask = syn.host_image_rays(H, W, K, R, T, body["can_bounds"])
    assert mask.all()
    pick = np.linspace(0, H * W - 1, N_BENCH_RAYS).astype(np.int64)
    batch = syn.make_batch(body, ray_o[pick], ray_d[pick], near[pick], far[pick], np.ones(len(pick), bool))
    return r, sd, body, batch, pick


# A batch of TWO frames (B = 2; lib/config/config.py:81 defaults train.batch_size to 4, every shipped YAML sets 1).  The
# reference's own Network cannot run B > 1 (latent_xyzc.py:35-36 pairs 6890 feature rows with B * 6890 coordinates), so the
# defined semantics — and the fixture, make_golden.run_batch2 — are its B = 1 results frame by frame, each frame encoded at the
# batch's common out_sh (if_clight_renderer.py:40-41: the maximum over the batch).
BATCH2 = dict(weights_seed=2, num_train_frame=7, n_samples=64, n_rays=256, mode="train", weights_kw=dict(alpha_bias=0.0, alpha_scale=12.0),
              frames=[dict(body=_SMALL_BODY, cam=_SMALL_CAM, latent_index=3),
                      dict(body=dict(seed=5, box=(0.25, 0.32, 0.2), rh=(-0.2, 0.1, 0.3), th=(-0.1, 0.15, 0.05)),
                           cam=dict(H=32, W=32, focal_factor=3.0, distance=1.4), latent_index=5)])


def build_batch2():
    """-> (recipe, state_dict_np, batch_np with leading dimension 2, [per-frame batch_np at the common out_sh])"""
    r = BATCH2
    sd = syn.make_weights(r["weights_seed"], num_train_frame=r["num_train_frame"], **r["weights_kw"])
    frames = []
    for f in r["frames"]:
        body = syn.make_body(**f["body"])
        c = f["cam"]
        K, R, T = syn.make_camera(body, c["H"], c["W"], focal_factor=c["focal_factor"], distance=c["distance"])
        ro, rd, near, far, mask = syn.host_image_rays(c["H"], c["W"], K, R, T, body["can_bounds"])
        assert ro.shape[0] >= r["n_rays"], ro.shape
        pick = np.linspace(0, ro.shape[0] - 1, r["n_rays"]).astype(np.int64)
        frames.append(syn.make_batch(body, ro[pick], rd[pick], near[pick], far[pick], np.ones(len(pick), bool),
                                     latent_index=f["latent_index"]))
    out_sh = np.max(np.concatenate([f["out_sh"] for f in frames], 0), 0, keepdims=True)
    assert not all(np.array_equal(f["out_sh"], out_sh) for f in frames), "the frames should differ in out_sh"
    batch = {k: np.concatenate([f[k] for f in frames], 0) for k in frames[0]}
    for f in frames:
        f["out_sh"] = out_sh.copy()
    return r, sd, batch, frames
