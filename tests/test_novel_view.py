"""Novel-view driving: host camera / pose algebra against fixtures produced by the
unmodified reference (tests/golden/make_golden.py::run_novel), the view-sharding driver under gloo (world size 2),
and — on the GPU — nb_image_assemble and the raygen -> render -> assemble loop against the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neuralbody_amd import novel_view as nv
from tests.golden import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def fix():
    return np.load(os.path.join(GOLDEN, "novel_view.npz"))


def test_load_cam_and_gen_path_match_reference(fix):
    r, body, cams, center = scenes.build_novel()
    K, RT = nv.load_cam(cams, r["ratio"])
    assert np.array_equal(np.array(K), fix["load_cam/K"]) and np.array_equal(np.array(RT), fix["load_cam/RT"])
    keep = [m.copy() for m in RT]
    for name, c in (("auto", None), ("center", center)):
        path = np.array(nv.gen_path(RT, r["num_render_views"], center=c))
        assert path.shape == fix["gen_path/" + name].shape
        assert np.abs(path - fix["gen_path/" + name]).max() <= 1e-12, name
        # each entry is a rigid world-to-camera transform
        for m in path:
            assert np.allclose(m[:3, :3] @ m[:3, :3].T, np.eye(3), atol=1e-12) and np.allclose(m[3], [0, 0, 0, 1])
    assert all(np.array_equal(a, b) for a, b in zip(keep, RT)), "gen_path must not overwrite its input"


def test_rotate_smpl_frame_matches_reference_dataset(fix):
    r, body, cams, center = scenes.build_novel()
    ts = np.arange(0, np.pi * 2, np.pi / 72)  # monocular_demo_dataset.py:25
    for step in r["turntable_steps"]:
        f = nv.rotate_smpl_frame(body["world_verts"], fix["turntable/rvec"], body["Th"].reshape(3), ts[step])
        pre = "turntable/%d/" % step
        assert np.array_equal(f["coord"], fix[pre + "coord"]), "voxel coordinates are index work: bit-exact"
        assert np.array_equal(f["out_sh"], fix[pre + "out_sh"])
        for k in ("can_bounds", "bounds", "Th"):
            assert np.array_equal(f[k], fix[pre + k]), k
        assert np.abs(f["R"] - fix[pre + "R"]).max() <= 2e-7  # the reference's R -> Rh -> R round trip, fp32
        assert f["coord"].dtype == np.int32 and f["coord"].min() >= 0 and (f["coord"].max(0) < f["out_sh"]).all()


def test_view_assignment_covers_every_view_once():
    for n, w in ((144, 8), (7, 4), (3, 8), (0, 2)):
        got = sorted(v for r in range(w) for v in nv.view_assignment(n, r, w))
        assert got == list(range(n))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_view(v, H=6, W=5):
    return torch.full((H, W, 3), float(v)) + torch.arange(H * W * 3, dtype=torch.float32).view(H, W, 3) / 1000.0


def _worker(rank, world, port, n_views, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
    try:
        calls = []

        def render_view(v):
            calls.append(v)
            return _fake_view(v)

        imgs = nv.render_views_sharded(render_view, n_views, 6, 5, "cpu")
        assert calls == nv.view_assignment(n_views, rank, world), "a rank renders only its own views"
        assert len(imgs) == n_views and all(torch.equal(imgs[v], _fake_view(v)) for v in range(n_views))
        mine = nv.render_views_sharded(_fake_view, n_views, 6, 5, "cpu", gather=False)
        assert sorted(mine) == nv.view_assignment(n_views, rank, world)
        np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.ones(1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_views", [5, 4])
def test_render_views_sharded_gloo_world2(tmp_path, n_views):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_views, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok_%d.npy" % r)) for r in range(world))


def test_render_views_without_process_group():
    imgs = nv.render_views_sharded(_fake_view, 3, 6, 5, "cpu")
    assert len(imgs) == 3 and torch.equal(imgs[2], _fake_view(2))


# ------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_image_assemble_matches_numpy():
    from neuralbody_amd import ops

    dev = "cuda:0"
    rs = np.random.RandomState(0)
    for H, W, p in ((37, 53, 0.4), (64, 64, 1.0), (16, 16, 0.0), (300, 301, 0.7)):
        mask = rs.uniform(size=H * W) < p
        n = int(mask.sum())
        rgb, depth = rs.uniform(size=(n, 3)).astype(np.float32), rs.uniform(1, 3, size=n).astype(np.float32)
        for white, bgr, scale in ((False, False, 1.0), (True, True, 255.0)):
            ref = np.full((H * W, 3), 1.0 if white else 0.0, np.float32)  # if_nerf_demo.py:22-26
            ref[mask] = rgb
            if bgr:
                ref = ref[:, [2, 1, 0]]
            ref = ref * np.float32(scale)
            dref = np.zeros(H * W, np.float32)
            dref[mask] = depth
            img, d = ops.image_assemble(torch.from_numpy(mask).to(dev), torch.from_numpy(rgb).to(dev), torch.from_numpy(depth).to(dev),
                                        white_bkgd=white, bgr=bgr, scale=scale)
            assert np.array_equal(img.cpu().numpy(), ref) and np.array_equal(d.cpu().numpy(), dref)
    img, d = ops.image_assemble(torch.zeros(0, dtype=torch.uint8, device=dev), torch.zeros((0, 3), device=dev))
    assert img.shape == (0, 3) and d is None


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32", "f16f6"])
def test_novel_view_loop_matches_oracle(precision):
    """gen_path camera -> nb_raygen -> Renderer.render -> nb_image_assemble == oracle image_rays + render + scatter."""
    from oracle import neuralbody_oracle as orc
    from tests import helpers as H_

    dev = "cuda:0"
    r, sd, body, batch, cam, _ = scenes.build("small")
    Hh = Ww = 40
    K0, R0, T0 = scenes.syn.make_camera(body, Hh, Ww, focal_factor=1.6, distance=1.8)
    train_RT = []
    for yaw in (-0.6, 0.0, 0.5, 1.1):
        _, R, T = scenes.syn.make_camera(body, Hh, Ww, focal_factor=1.6, distance=1.8, yaw=yaw)
        train_RT.append(np.concatenate([np.concatenate([R, T.reshape(3, 1)], 1), [[0, 0, 0, 1.0]]], 0))
    path = nv.gen_path(train_RT, 5, center=body["world_verts"].mean(0).astype(np.float64))
    net = H_.make_network(sd, dev, True, precision)
    rend = H_.make_renderer(net, r)
    frame = {k: v for k, v in H_.device_batch(batch, dev).items() if k in ("coord", "out_sh", "bounds", "R", "Th", "latent_index")}
    nvr = nv.NovelViewRenderer(rend, Hh, Ww, dev)
    nvr_reuse = nv.NovelViewRenderer(rend, Hh, Ww, dev, reuse_volumes=True)
    sdt = orc.tensor_state_dict(sd)
    checked = 0
    for RT in path[:3]:
        with torch.no_grad():
            out = nvr.render_view(K0, RT, body["can_bounds"], frame)
        torch.cuda.synchronize()
        ro, rd, near, far, mask = orc.image_rays(Hh, Ww, K0, RT[:3, :3], RT[:3, 3:], body["can_bounds"])
        assert np.array_equal(out["mask_at_box"].cpu().numpy().reshape(-1).astype(bool), np.asarray(mask).reshape(-1))
        assert out["n_rays"] == int(np.asarray(mask).sum())
        if out["n_rays"] == 0:
            continue
        ob = dict(batch)
        ob.update(ray_o=np.asarray(ro)[None], ray_d=np.asarray(rd)[None], near=np.asarray(near)[None], far=np.asarray(far)[None])
        with torch.no_grad():
            ref = orc.render(sdt, ob, n_samples=r["n_samples"], training=True)
        img = np.zeros((Hh * Ww, 3), np.float32)
        img[np.asarray(mask).reshape(-1)] = ref["rgb_map"][0].numpy()
        err = np.abs(out["img"].cpu().numpy().reshape(-1, 3) - img).max()
        assert err <= H_.RGB_TOL, err
        # encoding the frame once and reusing the volumes gives the very same image
        again = nvr_reuse.render_view(K0, RT, body["can_bounds"], frame)
        assert torch.equal(again["img"], out["img"]) and torch.equal(again["depth"], out["depth"])
        checked += 1
    assert checked >= 2, "the spiral must see the body"
    # the look-ahead generator (ray generation one view ahead of the march) yields the per-call images, in order
    singles = [nvr.render_view(K0, RT, body["can_bounds"], frame) for RT in path]
    ahead = list(nvr.render_views((K0, RT, body["can_bounds"], frame) for RT in path))
    assert len(ahead) == len(singles) == 5
    for a, b in zip(ahead, singles):
        assert a["n_rays"] == b["n_rays"] and torch.equal(a["img"], b["img"]) and torch.equal(a["depth"], b["depth"])
        assert torch.equal(a["mask_at_box"], b["mask_at_box"])
    assert list(nvr.render_views(iter(()))) == []
    # the generator does not leak its no_grad into the consumer's loop body, nor past an abandoned iteration
    assert torch.is_grad_enabled()
    gen = nvr.render_views((K0, RT, body["can_bounds"], frame) for RT in path[:2])
    for _view in gen:
        assert torch.is_grad_enabled()
        break
    assert torch.is_grad_enabled()
    gen.close()
