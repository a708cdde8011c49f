"""Generate tests/golden/*.npz by running the UNMODIFIED reference (/root/reference)
on CPU through oracle/ref_harness.py.  Run from the repo root, in the build
container only:   python tests/golden/make_golden.py

Each fixture stores only OUTPUTS of the reference (plus a checksum of the seeded
inputs); the tests regenerate the inputs from the recipe in scenes.py.
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_harness as rh  # noqa: E402
from tests.golden import scenes  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def input_digest(sd, batch):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    for k in sorted(batch):
        h.update(k.encode())
        h.update(np.ascontiguousarray(batch[k]).tobytes())
    return h.hexdigest()


def run_scene(name):
    ns = rh.load()
    r, sd, body, batch, cam, t_rand = scenes.build(name)
    cfg = ns.cfg
    cfg.N_samples = r["n_samples"]
    cfg.white_bkgd = bool(r["white_bkgd"])
    cfg.perturb = 1.0 if r["perturb"] else 0.0
    cfg.raw_noise_std = 0.0
    net = rh.make_reference_network(sd, train_mode=(r["mode"] == "train"))
    ren = rh.make_reference_renderer(net)
    tb = rh.torch_batch(batch)
    real_rand = torch.rand
    if t_rand is not None:
        # if_clight_renderer.py:22 draws torch.rand(z_vals.shape); feed it a known tensor instead
        tr = torch.from_numpy(t_rand)
        torch.rand = lambda *a, **k: tr
    try:
        with torch.no_grad():
            out = ren.render(tb)
            # BN running statistics after exactly ONE forward (momentum 0.01, latent_xyzc.py:215)
            bn_after_one = {k: v.clone().numpy() for k, v in net.state_dict().items()
                            if k.endswith("running_mean") or k.endswith("running_var")}
            # explicit-point decode (latent_xyzc.py:91-126) for a subset of rays
            sp_input = ren.prepare_sp_input(tb)
            vols = net.encode_sparse_voxels(sp_input)
            sel = slice(0, None, scenes.RAW_RAY_STRIDE)
            wpts, z_vals = ren.get_sampling_points(tb["ray_o"][:, sel], tb["ray_d"][:, sel], tb["near"][:, sel],
                                                   tb["far"][:, sel]) if t_rand is None else (None, None)
            raw = None
            dens = None
            if wpts is not None:
                viewdir = tb["ray_d"][:, sel] / torch.norm(tb["ray_d"][:, sel], dim=2, keepdim=True)
                raw = ren.get_density_color(
                    wpts, viewdir, lambda x, v: net.calculate_density_color(x, v, vols, sp_input))
                dens = net.calculate_density(wpts.view(1, -1, 3), vols, sp_input)
    finally:
        torch.rand = real_rand
    g = {k: v.numpy() for k, v in out.items()}
    g["input_digest"] = np.array(input_digest(sd, batch))
    if raw is not None:
        g["raw_subset"] = raw.numpy()
        g["density_subset"] = dens.numpy()
    if r["probes"]:
        for li, v in enumerate(vols):
            v = v[0].permute(1, 2, 3, 0).reshape(-1, v.shape[1]).numpy()  # [DHW, C]
            active = np.abs(v).sum(1) > 0
            idx = scenes.probe_indices(active)
            g["vol%d_probe_idx" % li] = idx
            g["vol%d_probe_val" % li] = v[idx]
            g["vol%d_sum" % li] = np.array(v.astype(np.float64).sum())
            g["vol%d_abs_sum" % li] = np.array(np.abs(v.astype(np.float64)).sum())
            g["vol%d_nonzero_voxels" % li] = np.array(int(active.sum()))
            g["vol%d_shape" % li] = np.array(vols[li].shape)
    if r["mode"] == "train":
        for k, v in bn_after_one.items():
            g["bn/" + k] = v
    path = os.path.join(OUT, "scene_%s.npz" % name)
    np.savez_compressed(path, **g)
    print(name, "rays", out["rgb_map"].shape[1], "->", path, "%.0f KB" % (os.path.getsize(path) / 1024))


def run_bench(tag):
    """The BENCH scene at the headline size through the unmodified reference: scenes.N_BENCH_RAYS rays spread over pose 1 of the
    timed cycle, the encoder at the full out_sh (dense stand-in: ~1 min of CPU), train-mode BatchNorm as run.py renders.  The
    fixture holds the reference's maps for those rays + the last sample's density and the transmittance in front of it."""
    ns = rh.load()
    r, sd, body, batch, pick = scenes.build_bench(tag)
    cfg = ns.cfg
    cfg.N_samples = r["n_samples"]
    cfg.white_bkgd = False
    cfg.perturb = 0.0
    cfg.raw_noise_std = 0.0
    net = rh.make_reference_network(sd, train_mode=True)
    ren = rh.make_reference_renderer(net)
    tb = rh.torch_batch(batch)
    with torch.no_grad():
        out = ren.render(tb)
        sp_input = ren.prepare_sp_input(tb)
        vols = net.encode_sparse_voxels(sp_input)
        wpts, z_vals = ren.get_sampling_points(tb["ray_o"], tb["ray_d"], tb["near"], tb["far"])
        viewdir = tb["ray_d"] / torch.norm(tb["ray_d"], dim=2, keepdim=True)
        raw = ren.get_density_color(wpts, viewdir, lambda x, v: net.calculate_density_color(x, v, vols, sp_input))
    g = {k: v.numpy() for k, v in out.items()}
    raw = raw.numpy().reshape(len(pick), r["n_samples"], 4)
    g["sigma_last"] = raw[:, -1, 3]
    g["t_last"] = 1.0 - g["weights"][0][:, :-1].sum(1)
    g["pick"] = pick
    g["input_digest"] = np.array(input_digest(sd, batch))
    for li, v in enumerate(vols):
        g["vol%d_nonzero_voxels" % li] = np.array(int((v[0].abs().sum(0) > 0).sum()))
    path = os.path.join(OUT, "bench_%s.npz" % tag)
    np.savez_compressed(path, **g)
    print("bench", tag, "rays", out["rgb_map"].shape[1], "->", path, "%.0f KB" % (os.path.getsize(path) / 1024))


def run_batch2():
    """scenes.BATCH2: the reference renderer on each of the two frames (B = 1, the batch's common out_sh), outputs stacked —
    the B > 1 semantics (the reference itself indexes its 6890 codes out of bounds for B > 1, latent_xyzc.py:35-36)."""
    ns = rh.load()
    r, sd, batch, frames = scenes.build_batch2()
    cfg = ns.cfg
    cfg.N_samples = r["n_samples"]
    cfg.white_bkgd = False
    cfg.perturb = 0.0
    cfg.raw_noise_std = 0.0
    outs = []
    for f in frames:
        net = rh.make_reference_network(sd, train_mode=True)  # a fresh module per frame: the running statistics start alike
        ren = rh.make_reference_renderer(net)
        with torch.no_grad():
            outs.append(ren.render(rh.torch_batch(f)))
    g = {k: np.concatenate([o[k].numpy() for o in outs], 0) for k in outs[0]}
    g["input_digest"] = np.array(input_digest(sd, batch))
    try:  # what the unmodified reference does with the B = 2 batch itself: recorded, not asserted
        net = rh.make_reference_network(sd, train_mode=True)
        with torch.no_grad():
            rh.make_reference_renderer(net).render(rh.torch_batch(batch))
        g["reference_runs_b2"] = np.array(True)
    except Exception as e:  # noqa: BLE001
        g["reference_runs_b2"] = np.array(False)
        g["reference_b2_error"] = np.array("%s: %s" % (type(e).__name__, str(e)[:200]))
        print("the reference on the B = 2 batch itself:", g["reference_b2_error"])
    path = os.path.join(OUT, "scene_batch2.npz")
    np.savez_compressed(path, **g)
    print("batch2 ->", path, "%.0f KB" % (os.path.getsize(path) / 1024))


def run_raygen():
    """get_rays / get_near_far (if_nerf_data_utils.py:8-21,54-69) and image_rays
    (render_utils.py:120-137) on a non-square camera."""
    ns = rh.load()
    from tests import synthetic as syn

    g = {}
    for tag, body_kw, H, W, ff in (("a", dict(seed=3, box=(0.9, 1.7, 0.35), rh=(0.2, 0.4, 0.0), th=(0.3, 0.1, 0.2)), 40, 56, 1.1),
                                   ("b", dict(seed=4, box=(0.3, 0.5, 0.2)), 33, 17, 3.0)):
        body = syn.make_body(**body_kw)
        K, R, T = syn.make_camera(body, H, W, focal_factor=ff, distance=2.2, yaw=-0.6, pitch=0.25)
        ro, rd = ns.get_rays(H, W, K, R, T)
        ro32 = ro.reshape(-1, 3).astype(np.float32)
        rd32 = rd.reshape(-1, 3).astype(np.float32)
        near, far, mask = ns.get_near_far(body["can_bounds"], ro32, rd32.copy())
        ns.cfg.H, ns.cfg.W, ns.cfg.ratio = H, W, 1.0
        RT = np.concatenate([R, T], 1)
        iro, ird, inear, ifar, _c, _s, imask = ns.image_rays(RT, K, body["can_bounds"])
        assert np.array_equal(imask, mask)
        g.update({tag + "_ray_o": ro32[0], tag + "_ray_d": rd32.reshape(H, W, 3), tag + "_near": near.astype(np.float32),
                  tag + "_far": far.astype(np.float32), tag + "_mask": mask, tag + "_img_near": inear, tag + "_img_far": ifar,
                  tag + "_img_ray_d": ird})
    path = os.path.join(OUT, "raygen.npz")
    np.savez_compressed(path, **g)
    print("raygen ->", path, "%.0f KB" % (os.path.getsize(path) / 1024))


def run_masked(kind):
    """The reference's mask-culled renderers (lib/networks/renderer/if_clight_renderer_mmsk.py / _msk.py) on CPU."""
    import importlib

    ns = rh.load()
    r, sd, batch, (H, W) = scenes.build_masked(kind)
    cfg = ns.cfg
    cfg.N_samples, cfg.white_bkgd, cfg.perturb, cfg.raw_noise_std = r["n_samples"], False, 0.0, 0.0
    cfg.H, cfg.W, cfg.ratio = H, W, 1.0
    net = rh.make_reference_network(sd, train_mode=True)
    mod = importlib.import_module("lib.networks.renderer.if_clight_renderer_" + kind)
    ren = mod.Renderer(net)
    tb = rh.torch_batch(batch)
    with torch.no_grad():
        out = ren.render(tb)
        wpts, _ = ren.get_sampling_points(tb["ray_o"], tb["ray_d"], tb["near"], tb["far"])
        inside = ren.prepare_inside_pts(wpts, tb)
    g = {k: v.numpy() for k, v in out.items()}
    g["inside"] = inside.numpy().reshape(1, -1, r["n_samples"])
    g["input_digest"] = np.array(input_digest(sd, batch))
    path = os.path.join(OUT, "masked_%s.npz" % kind)
    np.savez_compressed(path, **g)
    print(kind, "rays", out["rgb_map"].shape[1], "inside fraction %.3f" % g["inside"].mean(), "rgb max %.3f" % g["rgb_map"].max(),
          "->", path, "%.0f KB" % (os.path.getsize(path) / 1024))


def run_mesh():
    """The reference's mesh renderer (lib/networks/renderer/if_mesh_renderer.py) on CPU.  mcubes / trimesh are absent
    from the image and are CPU post-processing outside the hot path: they are stubbed for the import, the fixture keeps
    the density cube the marching cubes would consume."""
    import importlib
    import types

    ns = rh.load()
    mc = types.ModuleType("mcubes")
    mc.marching_cubes = lambda cube, th: (np.zeros((0, 3)), np.zeros((0, 3), np.int64))
    sys.modules["mcubes"] = mc
    sys.modules["trimesh"].Trimesh = lambda v, t: ("mesh", len(v), len(t))
    r, sd, batch = scenes.build_mesh()
    net = rh.make_reference_network(sd, train_mode=True)
    mod = importlib.import_module("lib.networks.renderer.if_mesh_renderer")
    ren = mod.Renderer(net)
    with torch.no_grad():
        out = ren.render(rh.torch_batch(batch))
    cube = out["cube"]
    inside = batch["inside"][0].astype(bool)
    g = {"cube": cube.astype(np.float32), "n_inside": np.array(int(inside.sum())), "input_digest": np.array(input_digest(sd, batch))}
    assert np.array_equal(g["cube"].astype(np.float64), cube), "the cube holds float32 values"
    path = os.path.join(OUT, "mesh_cube.npz")
    np.savez_compressed(path, **g)
    print("mesh cube", cube.shape, "inside %d of %d" % (inside.sum(), inside.size), "alpha range %.2f..%.2f, > mesh_th=5: %d" %
          (cube.min(), cube.max(), (cube > 5).sum()), "->", path, "%.0f KB" % (os.path.getsize(path) / 1024))


def _cv2_rodrigues(x):
    """Stand-in for cv2.Rodrigues (cv2 is absent): vector -> matrix or matrix -> vector, returned as cv2 does in a tuple."""
    from neuralbody_amd.novel_view import rodrigues

    x = np.asarray(x, np.float64)
    if x.size == 3:
        return (rodrigues(x.reshape(3)), None)
    th = np.arccos(np.clip((np.trace(x) - 1) / 2, -1, 1))
    axis = np.array([x[2, 1] - x[1, 2], x[0, 2] - x[2, 0], x[1, 0] - x[0, 1]]) / (2 * np.sin(th))
    return ((axis * th).reshape(3, 1), None)


def run_novel():
    """Host camera / pose algebra of the novel-view loop, executed by the UNMODIFIED reference:
    render_utils.load_cam + gen_path (both `center` variants) and monocular_demo_dataset.Dataset.prepare_input."""
    import importlib
    import tempfile
    import types

    ns = rh.load()
    cfg = ns.cfg
    r, body, cams, center = scenes.build_novel()
    ru = importlib.import_module("lib.utils.render_utils")
    g = {}
    saved = (cfg.ratio, cfg.num_render_views, list(cfg.voxel_size))
    with tempfile.TemporaryDirectory() as tmp:
        ann = os.path.join(tmp, "annots.npy")
        np.save(ann, {"cams": cams}, allow_pickle=True)
        cfg.ratio, cfg.num_render_views = r["ratio"], r["num_render_views"]
        try:
            K, RT = ru.load_cam(ann)
            g["load_cam/K"], g["load_cam/RT"] = np.array(K), np.array(RT)
            g["gen_path/auto"] = np.array(ru.gen_path([m.copy() for m in RT]))
            g["gen_path/center"] = np.array(ru.gen_path([m.copy() for m in RT], center.copy()))
            # rotating-SMPL frames: the dataset class without its file-reading constructor
            for name in ("imageio", "plyfile"):
                sys.modules.setdefault(name, types.ModuleType(name))
            sys.modules["plyfile"].PlyData = object
            sys.modules["cv2"].Rodrigues = _cv2_rodrigues
            mod = importlib.import_module("lib.datasets.light_stage.monocular_demo_dataset")
            ds = object.__new__(mod.Dataset)
            ds.data_root = tmp
            ds.ts = np.arange(0, np.pi * 2, np.pi / 72)
            rvec = _cv2_rodrigues(body["R"])[0].reshape(3)
            ds.params = {"pose": [np.concatenate([rvec, np.zeros(69)])], "trans": [body["Th"].reshape(3).astype(np.float64)]}
            os.makedirs(os.path.join(tmp, "vertices"))
            np.save(os.path.join(tmp, "vertices", "0.npy"), body["world_verts"].astype(np.float32))
            g["turntable/rvec"] = rvec
            for step in r["turntable_steps"]:
                coord, out_sh, can_bounds, bounds, Rh, Th = ds.prepare_input(0, step)
                R = _cv2_rodrigues(Rh)[0].astype(np.float32)  # monocular_demo_dataset.py:125
                for k, v in (("coord", coord), ("out_sh", out_sh), ("can_bounds", can_bounds), ("bounds", bounds), ("R", R), ("Th", Th)):
                    g["turntable/%d/%s" % (step, k)] = v
        finally:
            cfg.ratio, cfg.num_render_views, cfg.voxel_size = saved
    path = os.path.join(OUT, "novel_view.npz")
    np.savez_compressed(path, **g)
    print("novel view:", {k: v.shape for k, v in g.items() if "turntable" not in k}, "->", path, "%.0f KB" % (os.path.getsize(path) / 1024))


def run_train_step():
    """One training step of the UNMODIFIED reference (NetworkWrapper, lib/train/trainers/if_nerf_clight.py:18-36) on CPU:
    loss and, for every parameter, the gradient's L2 norm, sum and a few probe entries."""
    ns = rh.load()
    r, sd, batch, t_rand = scenes.build_train()
    cfg = ns.cfg
    cfg.N_samples, cfg.white_bkgd, cfg.perturb, cfg.raw_noise_std = r["n_samples"], False, 1.0, 0.0
    net = rh.make_reference_network(sd, train_mode=True)
    wrapper = ns.NetworkWrapper(net)
    tb = rh.torch_batch(batch)
    real_rand = torch.rand
    tr = torch.from_numpy(t_rand)
    torch.rand = lambda *a, **k: tr  # if_clight_renderer.py:22
    try:
        ret, loss, stats, _ = wrapper(tb)
        loss.backward()
    finally:
        torch.rand = real_rand
    g = {"loss": np.array(float(loss)), "rgb_map": ret["rgb_map"].detach().numpy()}
    for name, p in net.named_parameters():
        gr = p.grad.detach().numpy().astype(np.float64)
        g["norm/" + name] = np.array(np.sqrt((gr ** 2).sum()))
        g["sum/" + name] = np.array(gr.sum())
        g["probe/" + name] = gr.reshape(-1)[scenes.grad_probe_indices(gr.shape)]
        g["max/" + name] = np.array(np.abs(gr).max())
    path = os.path.join(OUT, "train_step.npz")
    np.savez_compressed(path, **g)
    print("train step: loss %.6f ->" % float(loss), path, "%.0f KB" % (os.path.getsize(path) / 1024))


def run_trained():
    """TRAINED["steps"] optimisation steps of the UNMODIFIED reference (NetworkWrapper.forward = render + masked MSE,
    lib/train/trainers/if_nerf_clight.py:18-36; the loop body of lib/train/trainers/trainer.py:46-53: zero_grad, backward,
    clip_grad_value_(40), Adam step with the shipped lr 5e-4) on CPU, then the reference's render of the whole view in run.py's
    mode.  The fixture carries the optimised parameters, so the GPU tests can put the HIP path through weights that came out
    of an optimiser instead of an initialiser."""
    ns = rh.load()
    T = scenes.TRAINED
    r, sd, body, batch, cam, _ = scenes.build(T["base"])
    K, R, Tt, H, W = cam
    cfg = ns.cfg
    cfg.N_samples, cfg.white_bkgd, cfg.raw_noise_std = r["n_samples"], bool(r["white_bkgd"]), 0.0
    net = rh.make_reference_network(sd, train_mode=True)
    wrapper = ns.NetworkWrapper(net)
    params = [p for n, p in net.named_parameters() if n.startswith(scenes.TRAINED_PREFIXES)]
    assert sum(p.numel() for p in params) < 600000
    opt = torch.optim.Adam(params, lr=T["lr"])
    target = scenes.trained_target(H, W, batch["mask_at_box"][0])
    n_rays = batch["ray_o"].shape[1]
    rs = np.random.RandomState(T["seed"])
    gen = torch.Generator().manual_seed(T["seed"])
    real_rand = torch.rand
    losses = []
    cfg.perturb = 1.0
    try:
        torch.rand = lambda *a, **k: real_rand(*a, generator=gen, **k)  # if_clight_renderer.py:22, made reproducible
        for step in range(T["steps"]):
            pick = np.sort(rs.choice(n_rays, T["n_rand"], replace=False))
            b = dict(batch)
            for k in ("ray_o", "ray_d", "near", "far"):
                b[k] = batch[k][:, pick]
            b["mask_at_box"] = np.ones((1, T["n_rand"]), bool)
            b["rgb"] = target[pick][None]
            tb = rh.torch_batch(b)
            ret, loss, stats, _ = wrapper(tb)
            opt.zero_grad()
            loss = loss.mean()
            loss.backward()
            torch.nn.utils.clip_grad_value_(net.parameters(), 40)
            opt.step()
            losses.append(float(loss))
            if step % 25 == 0:
                print("  trained fixture: step %d loss %.5f" % (step, losses[-1]), flush=True)
    finally:
        torch.rand = real_rand
    cfg.perturb = 0.0
    ren = rh.make_reference_renderer(net)
    with torch.no_grad():
        out = ren.render(rh.torch_batch(batch))
    g = {k: v.numpy() for k, v in out.items()}
    g["loss_history"] = np.array(losses, np.float32)
    sd_after = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    moved = 0.0
    for k, v in sd_after.items():
        if k.startswith(scenes.TRAINED_PREFIXES):
            g["param/" + k] = v.astype(np.float32)
            moved = max(moved, float(np.abs(v - sd[k]).max()))
        elif not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            assert np.array_equal(v, sd[k]), "frozen parameter %s moved" % k
    path = os.path.join(OUT, "scene_small_trained.npz")
    np.savez_compressed(path, **g)
    print("trained: loss %.4f -> %.4f, largest parameter move %.3f ->" % (losses[0], np.mean(losses[-10:]), moved), path,
          "%.0f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    torch.manual_seed(0)
    names = sys.argv[1:] or list(scenes.SCENES) + ["raygen", "train", "mmsk", "msk", "mesh", "novel", "trained", "batch2"] + ["bench:" + t for t in scenes.BENCH]
    for n in names:
        if n == "raygen":
            run_raygen()
        elif n == "train":
            run_train_step()
        elif n == "trained":
            run_trained()
        elif n == "mesh":
            run_mesh()
        elif n == "novel":
            run_novel()
        elif n in ("mmsk", "msk"):
            run_masked(n)
        elif n == "batch2":
            run_batch2()
        elif n.startswith("bench:"):
            run_bench(n.split(":", 1)[1])
        else:
            run_scene(n)
