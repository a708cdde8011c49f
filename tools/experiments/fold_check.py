"""First-light checks of the fc_0-folded march (precision 'f16f6', nb_fold.hip + nb_march_fold.hip) on the GPU:

    python tools/experiments/fold_check.py [rows] [points] [small] [full] [time]
    NB_LIB_PATH=neuralbody_amd/lib/libnb_hip_tap.so python tools/experiments/fold_check.py tap     (build with NB_EXTRA_FLAGS="-DFOLD_TAP -include tools/experiments/fold_instrument.h")

rows    U rows of nb_fold_build against fp32 torch (V_rows @ fc_0[:, level]^T), index grids, the zero row, nb_sparsify
points  nb_decode_points f16f6 against f32: coherent lattice points (one pass), scattered points (sample groups), outside points
small   the 'small' fixture (32 x 32 rays far apart: single-sample groups) and a zoomed camera on it (one pass) against f32 / fixture
full    512 x 512 x 64 bench view: 4096 rays against the oracle (bench.parity_check)
time    march time on the bench view (HIP events)
tap     per-layer accumulators of workgroup 0, step 0 against the fp32 activation tap
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from neuralbody_amd import ops  # noqa: E402
from tests import synthetic as syn  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.golden import scenes  # noqa: E402

DEV = "cuda:0"
LEVEL_BASE = (0, 32, 96, 224)
LEVEL_C = (32, 64, 128, 128)


def small_setup(precision="f16f6"):
    r, sd, body, batch, cam, t_rand = scenes.build("small")
    net = H.make_network(sd, DEV, True, precision=precision)
    bd = H.device_batch(batch, DEV)
    rend = H.make_renderer(net, r)
    return r, sd, body, batch, net, bd, rend


def check_rows():
    r, sd, body, batch, net, bd, rend = small_setup()
    ok = True
    with torch.no_grad():
        sp = rend.prepare_sp_input(bd)
        vols = net.encode_sparse_voxels(sp)
        cl = [ops.volume_as_channels_last(v) for v in vols]
        fold, keep = net._fold_planes(vols, cl)
        urows = keep[0]
        torch.cuda.synchronize()
        w0 = net.fc_0.weight.detach()[:, :, 0]
        u = urows.view(torch.float16).float()
        assert float(u[fold.zero_row].abs().max()) == 0.0, "zero row"
        for l in range(4):
            grid, rows_lin, n_rows, cap = vols.sparse[l]
            n = int(n_rows)
            V = cl[l].reshape(-1, LEVEL_C[l])[rows_lin[:n].long()]
            ref = V @ w0[:, LEVEL_BASE[l]:LEVEL_BASE[l] + LEVEL_C[l]].T
            got = u[fold.row_base[l]:fold.row_base[l] + n]
            got = got[:, :256] + got[:, 256:]
            err = float((got - ref).abs().max())
            scale = float(ref.abs().max())
            # grid consistency: grid[rows_lin[r]] == r
            g = grid.reshape(-1)[rows_lin[:n].long()]
            gok = bool((g == torch.arange(n, device=DEV, dtype=torch.int32)).all())
            print("rows level %d: n %d cap %d  max|U - ref| %.3e (max |U| %.3f)  grid ok %s" % (l, n, cap, err, scale, gok))
            ok &= err <= 2e-6 * max(1.0, scale) and gok
        # sparsify of the dense volumes: its active set = non-zero voxels, a subset of the encoder's
        for l in range(4):
            g2, lin2, n2, cap2 = ops.sparsify(cl[l])
            torch.cuda.synchronize()
            nz = (cl[l].reshape(-1, LEVEL_C[l]) != 0).any(1)
            n2 = int(n2)
            lin_ref = torch.nonzero(nz).reshape(-1).int()
            same = n2 == lin_ref.numel() and bool((lin2[:n2] == lin_ref).all())
            gg = g2.reshape(-1)
            gok = bool((gg[nz] == torch.arange(n2, device=DEV, dtype=torch.int32)).all()) and bool((gg[~nz] == -1).all())
            print("sparsify level %d: %d non-zero voxels (encoder rows %d)  list ok %s  grid ok %s" % (l, n2, int(vols.sparse[l][2]), same, gok))
            ok &= same and gok
    print("ROWS", "OK" if ok else "FAILED")
    return ok


def check_points():
    r, sd, body, batch, net, bd, rend = small_setup()
    ok = True
    with torch.no_grad():
        sp = rend.prepare_sp_input(bd)
        vols = net.encode_sparse_voxels(sp)
        lb = net.latent_bias(sp["latent_index"])
        scene32 = net.make_scene(vols, sp)
        scenev = net.make_scene(vols, sp, "f16f6")
        verts = torch.from_numpy(body["world_verts"]).to(DEV)
        rs = np.random.RandomState(3)
        sets = {}
        # coherent: 64-point blocks of a 4 x 4 x 4 lattice with 3 mm pitch around random vertices
        lat = torch.stack(torch.meshgrid(*[torch.arange(4.0)] * 3, indexing="ij"), -1).reshape(-1, 3).to(DEV) * 0.003
        ctr = verts[torch.from_numpy(rs.choice(verts.shape[0], 40)).to(DEV)]
        sets["lattice (one pass)"] = (ctr[:, None] + lat[None]).reshape(-1, 3).contiguous()
        # 64 points along a 5 mm line: groups of 16
        line = torch.arange(64.0, device=DEV)[:, None] * torch.tensor([0.0, 0.0, 0.005], device=DEV)
        sets["lines (groups of 16)"] = (ctr[:8, None] - 0.1 * torch.tensor([0, 0, 1.0], device=DEV) + line[None]).reshape(-1, 3).contiguous()
        # scattered
        lo, hi = torch.from_numpy(body["can_bounds"][0]).to(DEV), torch.from_numpy(body["can_bounds"][1]).to(DEV)
        sc = lo + (hi - lo) * torch.rand(64 * 9 + 13, 3, device=DEV)
        sc[:5] += 3.0  # outside the volume
        sets["scattered (single samples)"] = sc.contiguous()
        for name, pts in sets.items():
            vd = torch.nn.functional.normalize(torch.randn_like(pts), dim=-1).contiguous()
            ref = ops.decode_points(scene32, net.packed_weights("f32"), lb, pts, vd, precision="f32")
            got = ops.decode_points(scenev, net.packed_weights("f16f6"), lb, pts, vd, precision="f16f6")
            dref = ops.decode_points(scene32, net.packed_weights("f32"), None, pts, None, density_only=True, precision="f32")
            dgot = ops.decode_points(scenev, net.packed_weights("f16f6"), None, pts, None, density_only=True, precision="f16f6")
            torch.cuda.synchronize()
            e = float((got - ref).abs().max())
            ed = float((dgot - dref).abs().max())
            nz = float((ref[:, 3] != ref[0, 3]).float().mean())
            print("points %-28s n %5d  raw max err f16f6 %.3e  density %.3e   |raw| max %.2f, varied %.2f" % (
                name, pts.shape[0], e, ed, float(ref.abs().max()), nz))
            ok &= e <= 2e-3 and ed <= 2e-3
    print("POINTS", "OK" if ok else "FAILED")
    return ok


def check_small():
    r, sd, body, batch, net, bd, rend = small_setup()
    ok = True
    with torch.no_grad():
        out = rend.render(bd)
        torch.cuda.synchronize()
        g = H.golden("small")
        e = float(np.abs(out["rgb_map"].cpu().numpy() - g["rgb_map"]).max())
        print("small fixture (rays far apart): rgb L-inf vs the reference fixture %.3e" % e)
        ok &= e <= H.RGB_TOL
        # a camera zoomed onto the body: neighbouring rays share voxels (one pass per step)
        Hh = Ww = 64
        K, R, T = syn.make_camera(body, Hh, Ww, focal_factor=40.0, distance=2.5)
        ro, rd, near, far, mask, n = ops.raygen(Hh, Ww, K, R, T, body["can_bounds"], DEV)
        n = int(n)
        print("zoomed camera: %d of %d pixels hit the box" % (n, Hh * Ww))
        b2 = dict(bd)
        b2.update(ray_o=ro[None, :n], ray_d=rd[None, :n], near=near[None, :n], far=far[None, :n], mask_at_box=mask[None].bool())
        from neuralbody_amd.renderer import RenderConfig, Renderer
        outs = {}
        for prec in ("f32", "f16f6"):
            netp = H.make_network(sd, DEV, True, precision=prec)
            rp = Renderer(netp, RenderConfig(N_samples=r["n_samples"], perturb=0.0, H=Hh, W=Ww))
            outs[prec] = rp.render(b2)
        torch.cuda.synchronize()
        for prec in ("f16f6",):
            e = float((outs[prec]["rgb_map"] - outs["f32"]["rgb_map"]).abs().max())
            ew = float((outs[prec]["weights"] - outs["f32"]["weights"]).abs().max())
            print("zoomed camera %s vs f32: rgb %.3e weights %.3e (acc mean %.3f)" % (prec, e, ew, float(outs["f32"]["acc_map"].mean())))
            ok &= e <= H.RGB_TOL
    print("SMALL", "OK" if ok else "FAILED")
    return ok


def check_full():
    import bench

    sd, body, net, rend, bd, n = bench.build_scene(DEV, precision="f16f6")
    with torch.no_grad():
        out = rend.render(bd)
        torch.cuda.synchronize()
        p = bench.parity_check(sd, net, rend, bd, 64)
    print("full 512x512x64 parity:", {k: (("%.3e" % v) if isinstance(v, float) else v) for k, v in p.items() if not isinstance(v, (list, dict))})
    ok = p.get("linf", 1.0) <= H.RGB_TOL
    print("FULL", "OK" if ok else "FAILED")
    return ok


def check_time():
    import bench

    res = {}
    for prec in ("f16f6", "f16f6"):
        sd, body, net, rend, bd, n = bench.build_scene(DEV, precision=prec)
        with torch.no_grad():
            for _ in range(3):
                rend.render(bd)
            ops.MARCH_EVENTS = []
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(10):
                rend.render(bd)
            t1.record()
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in ops.MARCH_EVENTS]
            ops.MARCH_EVENTS = None
        print("time %-7s march %.3f ms (min %.3f)  view %.3f ms" % (prec, float(np.mean(ms)), float(np.min(ms)), t0.elapsed_time(t1) / 10))
        res[prec] = float(np.mean(ms))
    return True


def check_tap():
    import bench

    sd, body, net, rend, bd, n = bench.build_scene(DEV, precision="f16f6")
    ok = True
    with torch.no_grad():
        sp = rend.prepare_sp_input(bd)
        vols = net.encode_sparse_voxels(sp)
        scenev = net.make_scene(vols, sp, "f16f6")
        scene32 = net.make_scene(vols, sp)
        lb = net.latent_bias(sp["latent_index"])
        ray_o, ray_d = bd["ray_o"][0].contiguous(), bd["ray_d"][0].contiguous()
        near, far = bd["near"][0].contiguous(), bd["far"][0].contiguous()
        order = rend._tile_order(bd, n, 0, n)
        S = 64
        t_vals = torch.linspace(0.0, 1.0, steps=S).to(DEV)
        out = ops.march(scenev, net.packed_weights("f16f6"), lb, ray_o, ray_d, near, far, t_vals, want_raw=True, precision="f16f6",
                        ray_order=order)
        torch.cuda.synchronize()
        tap = out["raw"].reshape(-1)[: (3 * 256 + 128) * 64].cpu().numpy()
        layers = [tap[i * 256 * 64:(i + 1) * 256 * 64].reshape(256, 64) for i in range(3)]
        view = tap[3 * 256 * 64:].reshape(128, 64)
        # workgroup 0 marches the rays of slots xcd_remap(0) * 64 ..: block 0 -> group 0
        idx = order[:64].long()
        z0 = near[idx] * (1.0 - t_vals[0]) + far[idx] * t_vals[0]
        pts = (ray_o[idx] + ray_d[idx] * z0[:, None]).contiguous()
        vd = (ray_d[idx] / ray_d[idx].norm(dim=-1, keepdim=True)).contiguous()
        raw32, dbg = ops.decode_points(scene32, net.packed_weights("f32"), lb, pts, vd, debug=True, precision="f32")
        dbg = dbg.cpu().numpy()
    names = ["fc_0 (h1)", "fc_1 (h2)", "fc_2 (h3)"]
    offs = [352, 608, 864]
    for i in range(3):
        ref = dbg[:, offs[i]:offs[i] + 256].T  # [feature, sample], post relu
        got = np.maximum(layers[i], 0.0)
        err = np.abs(got - ref)
        print("%-10s max |err| %.3e (ref max %.3e)  worst feature %d sample %d" % (
            names[i], err.max(), np.abs(ref).max(), *np.unravel_index(err.argmax(), err.shape)))
        if err.max() > 1e-3 * max(1.0, np.abs(ref).max()):
            ok = False
            bad_f = np.where(err.max(1) > 1e-3)[0]
            bad_s = np.where(err.max(0) > 1e-3)[0]
            print("   bad features (%d): %s" % (len(bad_f), bad_f[:40]))
            print("   bad samples  (%d): %s" % (len(bad_s), bad_s[:40]))
            print("   got[0:4, 0:8] =\n", got[:4, :8], "\n   ref[0:4, 0:8] =\n", ref[:4, :8])
    ref = dbg[:, 1376:1376 + 128].T
    got = np.maximum(view, 0.0)
    err = np.abs(got - ref)
    print("%-10s max |err| %.3e (ref max %.3e)" % ("view (V)", err.max(), np.abs(ref).max()))
    ok &= err.max() <= 1e-3 * max(1.0, np.abs(ref).max())
    print("TAP", "OK" if ok else "FAILED")
    return ok


if __name__ == "__main__":
    which = sys.argv[1:] or ["rows", "points", "small", "full", "time"]
    fns = {"rows": check_rows, "points": check_points, "small": check_small, "full": check_full, "time": check_time, "tap": check_tap}
    results = {}
    for w in which:
        try:
            results[w] = fns[w]()
        except Exception as e:  # keep going: one GPU call should answer as much as possible
            import traceback

            traceback.print_exc()
            results[w] = False
    print("SUMMARY", results)
