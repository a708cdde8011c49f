"""fc_0 folded into the volume (nb_fold_build, nb_sparsify, the voxel-list march of NB_PREC_F16F6) on the GPU: the planes
against fp32 torch products of the same volumes, the active-set detection, and the three ways a workgroup's voxel list is
marched (one pass, sample groups of 16, single samples) against the exact-fp32 kernel on the same points.  The end-to-end
parity of the arithmetic against the reference's fixtures is test_gpu_parity.py / test_gpu_fullsize.py."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import scenes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LEVEL_BASE = (0, 32, 96, 224)
LEVEL_C = (32, 64, 128, 128)


def _small(precision="f16f6", train=True):
    r, sd, body, batch, cam, _ = scenes.build("small")
    net = H.make_network(sd, DEV, train, precision=precision)
    bd = H.device_batch(batch, DEV)
    rend = H.make_renderer(net, r)
    return r, sd, body, net, bd, rend


def test_fold_rows_are_fc0_times_the_active_voxels():
    """urows[row_base[l] + r] = fc_0.weight[:, level l] . V_l[voxel of row r] as fp16 head + fp16 remainder (exact fp32 products:
    ~1e-6 relative), the index grids address the rows, the extra last row is zero."""
    from neuralbody_amd import ops

    r, sd, body, net, bd, rend = _small()
    with torch.no_grad():
        sp = rend.prepare_sp_input(bd)
        vols = net.encode_sparse_voxels(sp)
        cl = [ops.volume_as_channels_last(v) for v in vols]
        fold, keep = net._fold_planes(vols, cl)
        again, _ = net._fold_planes(vols, cl)
        assert again is fold, "the planes are built once per (volumes, fc_0 version)"
        urows = keep[0]
        torch.cuda.synchronize()
        w0 = net.fc_0.weight.detach()[:, :, 0]
        u = urows.view(torch.float16).float()
        assert u.shape[1] == 512 and float(u[fold.zero_row].abs().max()) == 0.0
        total = 0
        for l in range(4):
            grid, rows_lin, n_rows, cap = vols.sparse[l]
            n = int(n_rows)
            assert fold.row_base[l] == total and n <= cap
            total += cap
            V = cl[l].reshape(-1, LEVEL_C[l])[rows_lin[:n].long()]
            ref = V @ w0[:, LEVEL_BASE[l]:LEVEL_BASE[l] + LEVEL_C[l]].T
            got = u[fold.row_base[l]:fold.row_base[l] + n]
            got = got[:, :256] + got[:, 256:]
            err = float((got - ref).abs().max())
            assert err <= 2e-6 * max(1.0, float(ref.abs().max())), (l, err)
            assert float(ref.abs().max()) > 0.5
            g = grid.reshape(-1)[rows_lin[:n].long()]
            assert bool((g == torch.arange(n, device=DEV, dtype=torch.int32)).all())
            assert int((grid >= 0).sum()) == n
        assert fold.zero_row == total
        # a new fc_0 (an in-place update, as optimizer steps and load_state_dict make) rebuilds them
        net.fc_0.weight.mul_(1.5)
        other, _ = net._fold_planes(vols, cl)
        assert other is not fold


def test_sparsify_finds_the_nonzero_voxels():
    """nb_sparsify: the active set of a dense volume = its non-zero voxels in linear order, grid = their numbering; an all-zero
    volume has none."""
    from neuralbody_amd import ops

    r, sd, body, net, bd, rend = _small()
    with torch.no_grad():
        vols = net.encode_sparse_voxels(rend.prepare_sp_input(bd))
        for l, v in enumerate(vols):
            cl = ops.volume_as_channels_last(v)
            grid, rows_lin, n_rows, cap = ops.sparsify(cl)
            torch.cuda.synchronize()
            nz = (cl.reshape(-1, LEVEL_C[l]) != 0).any(1)
            n = int(n_rows)
            lin_ref = torch.nonzero(nz).reshape(-1).int()
            assert n == lin_ref.numel() and bool((rows_lin[:n] == lin_ref).all())
            gg = grid.reshape(-1)
            assert bool((gg[nz] == torch.arange(n, device=DEV, dtype=torch.int32)).all()) and bool((gg[~nz] == -1).all())
            assert n <= int(vols.sparse[l][2])  # a subset of the encoder's rows (a row may be all zero after the ReLU)
        empty = torch.zeros((4, 6, 5, 32), device=DEV)
        grid, rows_lin, n_rows, cap = ops.sparsify(empty)
        assert int(n_rows) == 0 and int((grid >= 0).sum()) == 0


def test_foreign_dense_volumes_decode_like_the_encoders():
    """Volumes handed over WITHOUT their index structures (a plain list, e.g. from another encoder) take nb_sparsify; the
    decoder output equals the one on the encoder's own FeatureVolumes up to the rows that are all zero."""
    r, sd, body, net, bd, rend = _small()
    with torch.no_grad():
        sp = rend.prepare_sp_input(bd)
        vols = net.encode_sparse_voxels(sp)
        wpts, _ = rend.get_sampling_points(bd["ray_o"][:, ::7], bd["ray_d"][:, ::7], bd["near"][:, ::7], bd["far"][:, ::7])
        vd = torch.nn.functional.normalize(bd["ray_d"][:, ::7], dim=2)[:, :, None].expand_as(wpts).reshape(1, -1, 3)
        w = wpts.reshape(1, -1, 3)
        a = net.calculate_density_color(w, vd, vols, sp)
        b = net.calculate_density_color(w, vd, [v.clone() for v in vols], sp)  # a plain list: no sparse structures
    torch.cuda.synchronize()
    assert H.same_bits(a, b)


@pytest.mark.parametrize("kind", ["lattice", "lines", "clusters", "scattered"])
def test_points_on_every_marching_tier_match_the_fp32_kernel(kind):
    """64 points per workgroup whose voxel list (a) fits one pass (a 4 x 4 x 4 lattice of 3 mm pitch), (b) is up to 1024 voxels long
    (a 5 mm line of 64 points: passes of 128), (c) is the union of four distant clusters of 16 points (groups of 16), (d) is
    scattered over the whole volume (single samples; some outside the volume): raw and density against the exact-fp32 kernel."""
    from neuralbody_amd import ops

    r, sd, body, net, bd, rend = _small()
    with torch.no_grad():
        sp = rend.prepare_sp_input(bd)
        vols = net.encode_sparse_voxels(sp)
        lb = net.latent_bias(sp["latent_index"])
        scene32 = net.make_scene(vols, sp)
        scene16 = net.make_scene(vols, sp, "f16f6")
        verts = torch.from_numpy(body["world_verts"]).to(DEV)
        rs = np.random.RandomState(3)
        ctr = verts[torch.from_numpy(rs.choice(verts.shape[0], 24)).to(DEV)]
        if kind == "lattice":
            lat = torch.stack(torch.meshgrid(*[torch.arange(4.0)] * 3, indexing="ij"), -1).reshape(-1, 3).to(DEV) * 0.003
            pts = (ctr[:, None] + lat[None]).reshape(-1, 3)
        elif kind == "lines":
            line = torch.arange(64.0, device=DEV)[:, None] * torch.tensor([0.0, 0.0, 0.005], device=DEV)
            pts = (ctr[:8, None] - torch.tensor([0, 0, 0.1], device=DEV) + line[None]).reshape(-1, 3)
        elif kind == "clusters":
            lat = torch.stack(torch.meshgrid(torch.arange(4.0), torch.arange(4.0), torch.arange(1.0), indexing="ij"), -1).reshape(-1, 3).to(DEV) * 0.004
            far = verts[torch.from_numpy(rs.choice(verts.shape[0], 96)).to(DEV)]  # 24 workgroups x 4 clusters, anywhere on the body
            pts = (far[:, None] + lat[None]).reshape(-1, 3)
        else:
            lo, hi = torch.from_numpy(body["can_bounds"][0]).to(DEV), torch.from_numpy(body["can_bounds"][1]).to(DEV)
            pts = lo + (hi - lo) * torch.rand(64 * 9 + 13, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
            pts[:5] += 3.0
        pts = pts.contiguous()
        vd = torch.nn.functional.normalize(torch.randn(pts.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2)), dim=-1).contiguous()
        ref = ops.decode_points(scene32, net.packed_weights("f32"), lb, pts, vd, precision="f32")
        got = ops.decode_points(scene16, net.packed_weights("f16f6"), lb, pts, vd, precision="f16f6")
        dref = ops.decode_points(scene32, net.packed_weights("f32"), None, pts, None, density_only=True, precision="f32")
        dgot = ops.decode_points(scene16, net.packed_weights("f16f6"), None, pts, None, density_only=True, precision="f16f6")
    torch.cuda.synchronize()
    assert float(ref.abs().max()) > 5 and float((ref[:, 3] != ref[0, 3]).float().mean()) > 0.9, "degenerate points"
    # raw logits reach |20| (alpha_fc x20 / rgb_fc x8 gains): the tolerance of test_decode_points_stages_against_oracle
    # logits / densities of |20| with the synthetic gains: the four-bit cross terms move them by ~2e-3 (the RGB contract is 1e-4 on
    # composited rays, held by the render tests)
    H.assert_close(got.cpu().numpy(), ref.cpu().numpy(), 2.5e-3, "raw %s" % kind)
    H.assert_close(dgot.cpu().numpy(), dref.cpu().numpy(), 2.5e-3, "density %s" % kind)


def test_sparsify_beyond_the_two_kernel_scan():
    """More than 4 Mi voxels: nb_exclusive_scan takes its three-kernel form (the block totals get their own pass); a handful of
    non-zero voxels at known places, in linear order."""
    from neuralbody_amd import ops

    D, H, W, C = 66, 256, 256, 32  # 4.3 Mi voxels
    vol = torch.zeros((D, H, W, C), device=DEV)
    lin = torch.tensor([0, 1023, 1024, 1 << 20, (1 << 22) + 5, D * H * W - 1], device=DEV)
    vol.view(-1, C)[lin, 7] = 1.0
    grid, rows_lin, n_rows, cap = ops.sparsify(vol)
    torch.cuda.synchronize()
    assert int(n_rows) == lin.numel() and bool((rows_lin[:lin.numel()].long() == lin).all())
    g = grid.reshape(-1)
    assert bool((g[lin] == torch.arange(lin.numel(), device=DEV, dtype=torch.int32)).all()) and int((g >= 0).sum()) == lin.numel()


def test_dense_volumes_are_made_on_first_access_only(monkeypatch):
    """Inference: encode_sparse_voxels returns the levels' active rows; the march of the default arithmetic renders from them
    without a dense volume ever existing, bit for bit what the eagerly scattered volumes render; looking at the list
    materialises `.dense()` (the reference's [1,C,D,H,W] tensors), equal to the eager ones."""
    from neuralbody_amd import network as nbnet

    r, sd, body, net, bd, rend = _small(train=False)
    with torch.no_grad():
        sp = rend.prepare_sp_input(bd)
        lazy = net.encode_sparse_voxels(sp)
        assert isinstance(lazy, nbnet.FeatureVolumes) and not lazy.is_dense() and len(lazy) == 4
        out_lazy = rend.render(bd, feature_volume=lazy)
        assert not lazy.is_dense(), "the f16f6 march must not have materialised the dense volumes"
        assert net.fold_saturated(lazy) == 0
        monkeypatch.setattr(nbnet, "LAZY_DENSE", False)
        eager = net.encode_sparse_voxels(sp)
        assert eager.is_dense()
        out_eager = rend.render(bd, feature_volume=eager)
        for k in ("rgb_map", "acc_map", "depth_map", "weights"):
            assert H.same_bits(out_lazy[k], out_eager[k]), k
        for l in range(4):
            assert tuple(lazy[l].shape) == tuple(eager[l].shape) and H.same_bits(lazy[l].contiguous(), eager[l].contiguous()), l
        assert lazy.is_dense()
        # the exact arithmetic reads the dense volumes: it materialises them itself
        net32 = H.make_network(sd, DEV, False, precision="f32")
        rend32 = H.make_renderer(net32, r)
        monkeypatch.setattr(nbnet, "LAZY_DENSE", True)
        lz = net32.encode_sparse_voxels(sp)
        assert not lz.is_dense()
        a = rend32.render(bd, feature_volume=lz)["rgb_map"]
        assert lz.is_dense()
        monkeypatch.setattr(nbnet, "LAZY_DENSE", False)
        b = rend32.render(bd)["rgb_map"]
        assert H.same_bits(a, b)


def test_planes_that_leave_the_fp16_range_are_counted_and_auto_takes_the_exact_kernel():
    """nb_fold_build counts the fc_0 . V products beyond +-65504: 0 for the fixture's weights; with fc_0 scaled by
    1e6 the count is not 0, `auto` warns once and renders with the exact kernel — the same pixels as precision 'f32'."""
    r, sd, body, net, bd, rend = _small(precision="auto", train=False)
    with torch.no_grad():
        rend.render(bd)
        assert net.march_precision() == "f16f6"
        net.fc_0.weight.mul_(1e6)
        net.fc_0.bias.mul_(1e6)
        with pytest.warns(UserWarning, match="exceed the fp16 range"):
            out = rend.render(bd)
        assert net.march_precision() == "f32"
        net32 = H.make_network(sd, DEV, False, precision="f32")
        net32.fc_0.weight.mul_(1e6)
        net32.fc_0.bias.mul_(1e6)
        ref = H.make_renderer(net32, r).render(bd)
    assert H.same_bits(out["rgb_map"], ref["rgb_map"])


def test_a_later_frame_that_saturates_is_noticed_at_the_next_host_synchronisation():
    """saturation depends on the frame's volumes as well as on fc_0, and 'auto' reads the counter only once per weight
    version (a read-back per frame would drain the launch queue).  A later frame's counter is parked and read where the host waits
    for the device anyway — the renderer's per-frame out_sh read-back: the frame that saturates is marched clamped once, the next
    frame warns and runs on the exact kernel."""
    r, sd, body, net, bd, rend = _small(precision="auto", train=False)
    with torch.no_grad():
        rend.render(bd)  # first frame of these weights: counter read, 0
        assert net.march_precision() == "f16f6" and net._sat_pending is None
        # "a later frame with larger volumes": the volumes leave the encoder through a BatchNorm, so the test scales the last
        # BatchNorm of the coarsest level (the check is keyed on fc_0 alone, which does not change)
        bn = net.xyzc_net.conv4[7]
        bn.weight.mul_(1e7)
        bn.bias.mul_(1e7)
        frame2 = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in bd.items()}  # fresh tensors: a DataLoader's next batch
        rend.render(frame2)
        assert net._sat_pending is not None and net.march_precision() == "f16f6"  # parked, not read
        torch.cuda.synchronize()
        frame3 = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in bd.items()}
        with pytest.warns(UserWarning, match="exceed the fp16 range"):
            rend.render(frame3)  # its out_sh read-back looks at frame 2's counter first
        assert net.march_precision() == "f32"


def test_planes_of_foreign_volumes_are_sized_by_their_rows_and_kept():
    """A plain list of dense volumes: the planes hold as many rows as the volumes have non-zero voxels (not one per
    voxel of the grid), and a second decode of the same tensors reuses them."""
    r, sd, body, net, bd, rend = _small(train=False)
    with torch.no_grad():
        sp = rend.prepare_sp_input(bd)
        plain = [v.clone() for v in net.encode_sparse_voxels(sp)]
        wpts, _ = rend.get_sampling_points(bd["ray_o"][:, ::9], bd["ray_d"][:, ::9], bd["near"][:, ::9], bd["far"][:, ::9])
        w = wpts.reshape(1, -1, 3)
        a = net.calculate_density(w, plain, sp)
        first = net._foreign_fold
        nz = sum(int((v[0] != 0).any(0).sum()) for v in plain)
        assert first[3][1][0].shape[0] == first[3][0].zero_row + 1 and first[3][0].zero_row <= nz + 4
        b = net.calculate_density(w, plain, sp)
        assert net._foreign_fold is first and H.same_bits(a, b)
        plain[0].mul_(2.0)  # rewritten in place: a new version, new planes
        net.calculate_density(w, plain, sp)
        assert net._foreign_fold is not first
