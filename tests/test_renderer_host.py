"""Host-side logic of Renderer that needs no GPU: the ray tile order, the out_sh read-back cache, and the refusals of what the
HIP path does not implement (each must raise, with a message that says why, before any device work)."""
import pytest
import torch

from neuralbody_amd import ops
from neuralbody_amd.network import Network
from neuralbody_amd.renderer import RenderConfig, Renderer


def _renderer(H, W):
    return Renderer(Network(num_train_frame=3), RenderConfig(N_samples=8, perturb=0.0, H=H, W=W))


def _check_slots(order, mask, H, W, b, e):
    """A slot list of ops.tile_slots: every ray of [b, e) exactly once as a non-negative entry; 64 consecutive slots = one 8 x 8
    tile (16: a 4 x 4 block); padding slots reference a ray of their own tile; dead groups are entirely dead."""
    from neuralbody_amd._lib import SLOT_DEAD

    order = order.long()
    assert order.numel() == ((H + 7) // 8) * ((W + 7) // 8) * 64
    pix = torch.nonzero(mask.reshape(-1)).reshape(-1)[b:e]  # pixel of every ray of the range
    real = order[order >= 0]
    assert torch.equal(torch.sort(real).values, torch.arange(e - b))
    for g in range(order.numel() // 64):
        grp = order[64 * g:64 * g + 64]
        if (grp == SLOT_DEAD).any():
            assert (grp == SLOT_DEAD).all()
            continue
        rays = torch.where(grp >= 0, grp, -grp - 1)
        py, px = pix[rays] // W, pix[rays] % W
        assert int(py.max() - py.min()) <= 7 and int(px.max() - px.min()) <= 7 and int(py.min()) // 8 == int(py.max()) // 8
        assert (grp >= 0).any() and set((-grp[grp < 0] - 1).tolist()) <= set(grp[grp >= 0].tolist())
        for q in range(4):
            blk = grp[16 * q:16 * q + 16]
            r = blk[blk >= 0]
            if r.numel():
                assert int((pix[r] // W).max() - (pix[r] // W).min()) <= 3 and int((pix[r] % W).max() - (pix[r] % W).min()) <= 3


def test_tile_order_of_a_full_coverage_view_needs_no_mask_readback():
    """Every pixel a ray: the slot list is a function of the image geometry and the ray range; computed once per (geometry,
    range); 64 consecutive slots hold one 8 x 8 pixel tile; a rank's share of the rays leaves the other tiles dead."""
    H, W = 16, 24
    r = _renderer(H, W)
    n = H * W
    mask = torch.ones(n, dtype=torch.bool)
    full = r._tile_order({"mask_at_box": mask[None]}, n, 0, n)
    assert full.dtype == torch.int32 and int((full < 0).sum()) == 0
    _check_slots(full, mask, H, W, 0, n)
    again = r._tile_order({"mask_at_box": torch.ones(1, n, dtype=torch.bool)}, n, 0, n)  # another mask tensor, same geometry
    assert again is full
    part = r._tile_order({"mask_at_box": torch.ones(1, n, dtype=torch.bool)}, n, 64, 320)  # a rank's share of the rays
    assert part is not full
    _check_slots(part, mask, H, W, 64, 320)


def test_tile_order_of_a_partial_mask_follows_the_mask():
    """Pixels without a ray become padding slots of their tile (the tile's rays stay one workgroup), tiles without a ray are
    dead; image sizes that are no multiple of 8 are padded the same way."""
    H, W = 19, 27
    r = _renderer(H, W)
    mask = torch.zeros(H, W, dtype=torch.bool)
    mask[2:14, 3:20] = True
    mask[5, 7] = False
    n = int(mask.sum())
    batch = {"mask_at_box": mask.reshape(1, -1)}
    order = r._tile_order(batch, n, 0, n)
    _check_slots(order, mask, H, W, 0, n)
    assert r._tile_order(batch, n, 0, n) is order          # same tensor object, same version: cached
    assert r._tile_order(batch, n, 5, 5) is None           # an empty range (a rank without rays): nothing to order
    _check_slots(r._tile_order(batch, n, 0, 40), mask, H, W, 0, 40)  # fewer than 64 rays keep their tiles too (a rank's small share)
    _check_slots(r._tile_order(batch, n, 17, 150), mask, H, W, 17, 150)


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


# ------------------------------------------------------------------ refusals (one test per NotImplementedError)
def _batch(n_batch=1, n=64):
    z = lambda *sh: torch.zeros(*sh)  # noqa: E731
    return {"ray_o": z(n_batch, n, 3), "ray_d": torch.ones(n_batch, n, 3), "near": z(n_batch, n), "far": torch.ones(n_batch, n),
            "coord": torch.zeros(n_batch, 10, 3, dtype=torch.int32), "out_sh": torch.tensor([[32, 32, 32]] * n_batch, dtype=torch.int32),
            "bounds": z(n_batch, 2, 3), "R": torch.eye(3)[None].repeat(n_batch, 1, 1), "Th": z(n_batch, 1, 3),
            "latent_index": torch.zeros(n_batch, dtype=torch.long)}


def test_other_encoding_resolutions_are_refused():
    with pytest.raises(NotImplementedError, match="xyz_res=10, view_res=4"):
        Network(num_train_frame=3, xyz_res=8)
    with pytest.raises(NotImplementedError, match="latent_xyzc.py:27"):
        Network(num_train_frame=3, view_res=6)
    with pytest.raises(ValueError, match="precision must be"):
        Network(num_train_frame=3, precision="fp8")


def test_a_batch_of_frames_is_split_frame_by_frame():
    """B > 1 (lib/config/config.py:81 defaults train.batch_size to 4; the reference's own Network cannot run it: latent_xyzc.py:35-36
    pairs 6890 feature rows with B * 6890 coordinates) = B passes of one frame each.  Host side: `frame_sp_input` cuts sp_input as
    if_clight_renderer.py:29-52 built it back into frames; `BatchedFeatureVolumes` presents the frames' volumes as the reference's
    list of [B,C,D,H,W] tensors; make_scene describes one frame and says so."""
    from neuralbody_amd.network import BatchedFeatureVolumes, FeatureVolumes, frame_sp_input, frame_volumes

    r = _renderer(8, 8)
    B, n = 3, 5
    coord = torch.arange(B * n * 3, dtype=torch.int32).view(B, n, 3)
    batch = {"coord": coord, "out_sh": torch.tensor([[8, 16, 8], [16, 8, 8], [8, 8, 8]]), "bounds": torch.arange(B * 6.0).view(B, 2, 3),
             "R": torch.eye(3)[None].repeat(B, 1, 1) * torch.arange(1.0, B + 1)[:, None, None], "Th": torch.arange(B * 3.0).view(B, 1, 3),
             "latent_index": torch.arange(B)}
    sp = r.prepare_sp_input(batch)
    assert sp["coord"].shape == (B * n, 4) and sp["batch_size"] == B and sp["out_sh"] == [16, 16, 8]
    for b in range(B):
        f = frame_sp_input(sp, b)
        assert f["batch_size"] == 1 and f["out_sh"] == [16, 16, 8]
        assert torch.equal(f["coord"][:, 1:], coord[b]) and int(f["coord"][:, 0].abs().sum()) == 0
        assert torch.equal(f["R"], batch["R"][b:b + 1]) and torch.equal(f["Th"], batch["Th"][b:b + 1])
        assert torch.equal(f["bounds"], batch["bounds"][b:b + 1]) and torch.equal(f["latent_index"], batch["latent_index"][b:b + 1])
    frames = [FeatureVolumes([torch.full((1, c, 2, 2, 2), float(b)) for c in (32, 64, 128, 128)]) for b in range(B)]
    fv = BatchedFeatureVolumes(frames)
    assert len(fv) == 4 and [tuple(v.shape) for v in fv] == [(B, c, 2, 2, 2) for c in (32, 64, 128, 128)]
    assert float(fv[2][1].mean()) == 1.0 and frame_volumes(fv, 2) is frames[2]
    plain = list(fv)
    assert [tuple(v.shape) for v in frame_volumes(plain, 1)] == [(1, c, 2, 2, 2) for c in (32, 64, 128, 128)]
    with pytest.raises(ValueError, match="ONE frame"):
        r.net.make_scene([v[0].permute(1, 2, 3, 0).contiguous() for v in frames[0]], sp)
    with pytest.raises(ValueError, match="equal row counts"):
        r.net.encode_sparse_voxels({"coord": torch.zeros(21, 4, dtype=torch.int32), "out_sh": [32, 32, 32], "batch_size": 2})


def test_the_differentiable_path_refuses_what_it_does_not_differentiate():
    """Training goes through training.RenderFunction (one autograd.Function around the whole render); inference-only options
    are refused up front instead of silently returning tensors without gradients."""
    r = _renderer(8, 8)
    b = _batch()
    assert torch.is_grad_enabled() and any(p.requires_grad for p in r.net.parameters())
    with pytest.raises(NotImplementedError, match="inference-only"):
        r.render(b, want_raw=True)
    with pytest.raises(NotImplementedError, match="inference-only"):
        r.render(b, feature_volume=[None])


def test_prepare_sp_input_keeps_the_reference_layout():
    """if_clight_renderer.py:29-52: coord [B * n, 4] = (batch index, d, h, w); batch size 1 takes the constant-column path and
    hands the encoder the [n, 3] coordinates themselves, batch size 2 the general one — same values as the reference's recipe."""
    r = _renderer(16, 16)
    for B in (1, 2):
        coord = torch.arange(B * 5 * 3, dtype=torch.int32).reshape(B, 5, 3)
        batch = {"coord": coord, "out_sh": torch.tensor([[8, 8, 8]] * B, dtype=torch.int32), "bounds": torch.zeros(B, 2, 3),
                 "R": torch.eye(3)[None].repeat(B, 1, 1), "Th": torch.zeros(B, 1, 3), "latent_index": torch.zeros(B, dtype=torch.long)}
        sp = r.prepare_sp_input(batch)
        idx = torch.cat([torch.full([5], i, dtype=torch.int32) for i in range(B)])
        assert torch.equal(sp["coord"], torch.cat([idx[:, None], coord.view(-1, 3)], dim=1)) and sp["coord"].dtype == torch.int32
        assert sp["batch_size"] == B and sp["out_sh"] == [8, 8, 8]
        assert ("_coord_dhw" in sp) == (B == 1)
        if B == 1:
            assert sp["_coord_dhw"].data_ptr() == coord.data_ptr()


def test_prefetch_is_an_inference_call_on_device_tensors():
    r = _renderer(16, 16)
    batch = {"coord": torch.zeros(1, 5, 3, dtype=torch.int32)}
    with pytest.raises(RuntimeError, match="inference-only"):
        r.prefetch(batch)
    with torch.no_grad(), pytest.raises(RuntimeError, match="device tensors"):
        r.prefetch(batch)


def test_a_ticket_belongs_to_the_frame_tensors_it_was_made_from():
    """Renderer._ticket_is_for: identity AND version counters of coord, out_sh, bounds, R, Th, latent_index (+ the frame token);
    the rays may differ."""
    r = _renderer(16, 16)

    def frame():
        return {"coord": torch.zeros(1, 5, 3, dtype=torch.int32), "out_sh": torch.tensor([[8, 8, 8]], dtype=torch.int32),
                "bounds": torch.zeros(1, 2, 3), "R": torch.eye(3)[None], "Th": torch.zeros(1, 1, 3),
                "latent_index": torch.zeros(1, dtype=torch.long)}

    a = frame()
    ticket = (r._frame_key(a), None, None, None)
    view = dict(a, ray_o=torch.zeros(1, 4, 3))  # another view of the same frame: same tensor objects
    assert r._ticket_is_for(ticket, view)
    assert not r._ticket_is_for(ticket, frame())  # equal values, other tensors
    assert not r._ticket_is_for(ticket, dict(a, frame_token=3))
    a["Th"].add_(1.0)  # rewritten in place after the ticket was made
    assert not r._ticket_is_for(ticket, view)


def test_feature_volumes_materialise_on_first_access():
    """FeatureVolumes built from compact rows (what the inference encoder returns): four [1,C,D,H,W] tensors appear when the list
    is looked at — zeros at inactive voxels, rows beyond the device-side count ignored — and not before."""
    import torch

    from neuralbody_amd.network import FeatureVolumes

    g = torch.Generator().manual_seed(0)
    shapes, chans = [(4, 5, 3), (2, 3, 2), (2, 2, 2), (1, 2, 1)], (32, 64, 128, 128)
    sparse, rows, want = [], [], []
    for (D, H, W), c in zip(shapes, chans):
        nvox, cap = D * H * W, 7
        n = min(5, nvox)
        lin = torch.randperm(nvox, generator=g)[:n].sort().values.int()
        rows_lin = torch.zeros(cap, dtype=torch.int32)
        rows_lin[:n] = lin
        r = torch.randn(cap, c, generator=g)  # rows n.. are garbage the count must hide
        grid = torch.full((D, H, W), -1, dtype=torch.int32)
        grid.view(-1)[lin.long()] = torch.arange(n, dtype=torch.int32)
        sparse.append((grid, rows_lin, torch.tensor([n], dtype=torch.int32), cap))
        rows.append(r)
        d = torch.zeros(nvox, c)
        d[lin.long()] = r[:n]
        want.append(d.view(D, H, W, c).permute(3, 0, 1, 2)[None])
    fv = FeatureVolumes(None, sparse, rows=rows, shapes=shapes)
    assert len(fv) == 4 and not fv.is_dense() and bool(fv)
    assert not isinstance(fv, list)  # a C-level list consumer must not see "no volumes": it gets a TypeError instead
    with pytest.raises(TypeError):
        torch.cat(fv)
    assert not fv.is_dense()
    v2 = fv[2]
    assert fv.is_dense() and len(list(fv)) == 4 and len(fv[1:]) == 3
    assert v2.shape == (1, 128, 2, 2, 2)
    for got, ref in zip(fv, want):
        assert got.shape == ref.shape and torch.equal(got, ref)
    # eager volumes behave like the plain list they are
    eager = FeatureVolumes([w.clone() for w in want], sparse)
    assert eager.is_dense() and eager.shapes == [tuple(s) for s in shapes] and torch.equal(eager[0], want[0])


def test_zero_arena_hands_out_zeroed_aligned_views():
    import torch

    from neuralbody_amd import ops

    req = [((3,), torch.float32), ((2, 5), torch.float64), ((7, 16), torch.float32)]
    arena = ops.ZeroArena(ops.ZeroArena.size_of(req), "cpu")
    views = [arena.take(shape, dt) for shape, dt in req]
    for v, (shape, dt) in zip(views, req):
        assert tuple(v.shape) == tuple(shape) and v.dtype == dt and float(v.abs().sum()) == 0.0 and v.data_ptr() % 16 == 0
    views[0].fill_(1.0)
    assert float(views[1].abs().sum()) == 0.0 and float(views[2].abs().sum()) == 0.0  # the views do not overlap
    with pytest.raises(RuntimeError):
        arena.take((1,))


def test_sp_input_builds_the_reference_coord_layout_on_first_use():
    """prepare_sp_input (if_clight_renderer.py:29-52) for batch size 1: the [n, 4] = (batch index, d, h, w) tensor the reference's
    encode_sparse_voxels indexes is a concatenation launch per frame that this package's encoder never reads (it takes the [n, 3]
    coordinates, `_coord_dhw`) — `SpInput` builds it when somebody indexes 'coord', once, and is a dict otherwise."""
    from neuralbody_amd.renderer import SpInput

    r = _renderer(8, 8)
    b = _batch()
    sp = r.prepare_sp_input(b)
    assert isinstance(sp, SpInput) and isinstance(sp, dict)
    assert "coord" not in sp and sp.get("coord") is None and "_coord_dhw" in sp
    assert torch.equal(sp["_coord_dhw"], b["coord"].view(-1, 3))
    c = sp["coord"]  # the reference's access
    assert c.shape == (10, 4) and int(c[:, 0].abs().sum()) == 0 and torch.equal(c[:, 1:], b["coord"].view(-1, 3)) and c.dtype == b["coord"].dtype
    assert "coord" in sp and sp["coord"] is c  # built once
    cp = type(sp)(sp)  # a copy keeps the behaviour (the graph-captured prefetch copies its sp_input)
    assert isinstance(cp, SpInput) and cp["coord"] is c
    with pytest.raises(KeyError):
        sp["no such key"]
    for k in ("out_sh", "batch_size", "bounds", "R", "Th", "latent_index"):
        assert k in sp
