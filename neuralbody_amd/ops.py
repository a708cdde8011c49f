"""Tensor-level wrappers over the C ABI (include/nb_hip.h).  Every function validates device,
dtype, contiguity and shape, then hands raw device pointers + the current HIP stream to
libnb_hip.so.  Nothing here computes: there is no PyTorch / CPU fallback for any operator."""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import NbCull, NbError, NbFold, NbMlpParams, NbScene, check, ptr

LEVEL_CHANNELS = (32, 64, 128, 128)
DBG_WIDTH = 1600
# activation tap layout of nb_decode_points(debug=True): F | h1 | h2 | h3 | G | V | PE (csrc/nb_march_common.h TAP_*)
TAP = {"F": (0, 352), "h1": (352, 608), "h2": (608, 864), "h3": (864, 1120), "G": (1120, 1376), "V": (1376, 1504),
       "PE": (1504, 1594)}
# bench.py sets this to a list to collect (start, end) HIP events bracketing every nb_march launch on
# the stream it is enqueued on
MARCH_EVENTS = None
FIXUP_CAP = 32768  # rays the march's last-sample fix-up list holds by default (2 MB of scratch; the bench view lists ~50)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, dtype, shape=None, name="tensor"):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise NbError("%s must live on a HIP device (got %s); the HIP path has no CPU fallback" % (name, t.device))
    if t.dtype != dtype:
        raise ValueError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if shape is not None:
        if len(shape) != t.dim() or any(s is not None and s != d for s, d in zip(shape, t.shape)):
            raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), shape))
    return t


def volume_as_channels_last(v):
    """[1,C,D,H,W] (any strides) -> contiguous [D,H,W,C] view or copy."""
    if v.dim() != 5 or v.shape[0] != 1:
        raise ValueError("feature volume must be [1,C,D,H,W] (batch 1), got %s" % (tuple(v.shape),))
    return v[0].permute(1, 2, 3, 0).contiguous()  # no copy when the storage is already channels-last


def make_pose(R, Th, bounds, device=None):
    """The per-frame pose block the kernels read from DEVICE memory: R[9] row-major | Th[3] | bounds_min[3] (15 fp32).
    R (any shape with 9 elements), Th (>= 3 elements), bounds ([..,2,3] or 3 elements: the minimum corner is taken) may
    be device tensors — sp_input['R'|'Th'|'bounds'] as they are, no host round trip, no sync — or host sequences."""
    def dev_flat(x, n):
        if not isinstance(x, torch.Tensor):
            import numpy as np

            x = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))
        x = x.detach().reshape(-1)[:n].to(device=device if device is not None else x.device, dtype=torch.float32)
        if x.numel() != n:
            raise ValueError("pose component has %d elements, expected >= %d" % (x.numel(), n))
        return x

    if device is None:
        device = next((x.device for x in (R, Th, bounds) if isinstance(x, torch.Tensor)), None)
    pose = torch.cat([dev_flat(R, 9), dev_flat(Th, 3), dev_flat(bounds, 3)])
    return _req(pose, torch.float32, (15,), "pose")


def sparsify(volume_cl):
    """nb_sparsify: active set of a dense channels-last volume [D,H,W,C] that came without one -> (grid [D,H,W] int32,
    rows_lin [cap] int32, n_rows [1] int32, cap).  Capacity = every voxel (no device -> host read)."""
    _req(volume_cl, torch.float32, (None, None, None, None), "volume")
    dhw = [int(v) for v in volume_cl.shape[:3]]
    nvox = dhw[0] * dhw[1] * dhw[2]
    dev = volume_cl.device
    grid = torch.empty(dhw, dtype=torch.int32, device=dev)
    buf = torch.zeros(nvox + 1, dtype=torch.int32, device=dev)
    rows_lin, n_rows = buf[:nvox], buf[nvox:]
    scratch = scan_scratch(nvox, dev)
    check(_lib.lib().nb_sparsify(ptr(volume_cl), _i3(dhw), int(volume_cl.shape[3]), ptr(grid), ptr(rows_lin), ptr(n_rows),
                                 nvox, ptr(scratch), _stream()), "nb_sparsify")
    return grid, rows_lin, n_rows, nvox


def fold_build(volumes_cl, sparse, fc0_w, rows=None, n_sat=None):
    """nb_fold_build: the fc_0-folded planes of one frame (precision 'f16f6').  volumes_cl: the four channels-last volumes
    (or, with `rows`, just their [D,H,W] shapes); sparse: per level (grid, rows_lin, n_rows[1], n_rows_max) — the encoder's index
    structures, or `sparsify`'s; rows: per level the ACTIVE rows themselves, compact [>= n_rows_max, C] fp32 (the encoder's
    output before `.dense()`): the dense volumes are then not read and need not exist; fc0_w [256,352(,1)].
    Returns (NbFold, keepalive); keepalive[0] = the planes, keepalive[1] = n_saturated [1] int32 (see include/nb_hip.h)."""
    if fc0_w.dim() == 3:
        fc0_w = fc0_w[:, :, 0]
    fc0_w = fc0_w.detach()
    if not fc0_w.is_contiguous():
        fc0_w = fc0_w.contiguous()
    _req(fc0_w, torch.float32, (256, 352), "fc0_w")
    v4, l4, n4, caps = (C.c_void_p * 4)(), (C.c_void_p * 4)(), (C.c_void_p * 4)(), (C.c_int32 * 4)()
    f = NbFold()
    base = 0
    keep_src = []
    for l, (v, (grid, rows_lin, n_rows, cap)) in enumerate(zip(volumes_cl, sparse)):
        dhw = tuple(int(x) for x in (v.shape[:3] if isinstance(v, torch.Tensor) else v))
        _req(grid, torch.int32, dhw, "grid[%d]" % l)
        _req(n_rows, torch.int32, (1,), "n_rows[%d]" % l)
        cap = max(int(cap), 1)
        if rows is not None:
            r = _req(rows[l], torch.float32, (None, LEVEL_CHANNELS[l]), "rows[%d]" % l)
            if r.shape[0] < cap:
                raise ValueError("rows[%d] shorter than its capacity" % l)
            v4[l], l4[l] = r.data_ptr(), None
            keep_src.append(r)
        else:
            _req(v, torch.float32, (None, None, None, LEVEL_CHANNELS[l]), "volume[%d]" % l)
            _req(rows_lin, torch.int32, (None,), "rows_lin[%d]" % l)
            if rows_lin.shape[0] < cap:
                raise ValueError("rows_lin[%d] shorter than its capacity" % l)
            v4[l], l4[l] = v.data_ptr(), rows_lin.data_ptr()
            keep_src.append(v)
        n4[l], caps[l] = n_rows.data_ptr(), cap
        f.grid[l] = grid.data_ptr()
        f.row_base[l] = base
        base += cap
    if base >= (1 << 21):
        raise ValueError("fold_build: %d rows exceed the 2 GiB plane the march addresses with 32-bit offsets" % base)
    dev = fc0_w.device
    urows = torch.empty((base + 1, 512), dtype=torch.int16, device=dev)
    if n_sat is None:
        n_sat = torch.zeros(1, dtype=torch.int32, device=dev)
    _req(n_sat, torch.int32, (1,), "n_sat")
    f.urows = urows.data_ptr()
    f.zero_row = base
    check(_lib.lib().nb_fold_build(v4, l4, n4, caps, ptr(fc0_w), ptr(urows), ptr(n_sat), _stream()), "nb_fold_build")
    return f, [urows, n_sat, fc0_w] + keep_src + [t for sp in sparse for t in sp[:3]]


def make_scene(volumes_cl, pose, voxel_size, out_sh, fold=None):
    """volumes_cl: four contiguous [D,H,W,C] fp32 device tensors — or, with `fold` (precision 'f16f6' reads the folded planes,
    never the volumes), just their (D, H, W) shapes; pose: the 15-float DEVICE block of make_pose; voxel_size (3, dhw) and
    out_sh (3) are HOST sequences; fold: `fold_build`'s result.  Returns (NbScene, keepalive)."""
    sc = NbScene()
    if len(volumes_cl) != 4:
        raise ValueError("expected 4 feature volumes")
    keep = [pose]
    for l, v in enumerate(volumes_cl):
        if isinstance(v, torch.Tensor):
            _req(v, torch.float32, (None, None, None, LEVEL_CHANNELS[l]), "volume[%d]" % l)
            sc.vol[l] = v.data_ptr()
            keep.append(v)
            dhw = v.shape[:3]
        else:
            if fold is None:
                raise ValueError("volume[%d] given as a shape: only a scene with folded planes can do without the volume" % l)
            sc.vol[l] = None
            dhw = v
        for k in range(3):
            sc.vol_dhw[l][k] = int(dhw[k])
    _req(pose, torch.float32, (15,), "pose")
    sc.pose = pose.data_ptr()
    for k in range(3):
        sc.voxel_size[k] = float(voxel_size[k])
        sc.out_sh[k] = int(out_sh[k])
    if fold is not None:
        sc.fold = C.pointer(fold[0])
        keep += [fold[0]] + list(fold[1])
    return sc, keep


def mlp_pack_size():
    return int(_lib.lib().nb_mlp_pack_size())


def _mlp_params(params):
    """params: dict name -> fp32 device tensor ([out,in] or [out,in,1] weights, [out] biases)."""
    shapes = {"fc0": (256, 352), "fc1": (256, 256), "fc2": (256, 256), "alpha": (1, 256), "feature": (256, 256),
              "latent": (256, 384), "view": (128, 346), "rgb": (3, 128)}
    p = NbMlpParams()
    keep = []
    for name, (o, i) in shapes.items():
        w = params[name + "_w"]
        b = params[name + "_b"]
        if w.dim() == 3 and w.shape[2] == 1:
            w = w[:, :, 0]
        w = w.detach()
        b = b.detach()
        if not w.is_contiguous():
            w = w.contiguous()
        _req(w, torch.float32, (o, i), name + "_w")
        _req(b, torch.float32, (o,), name + "_b")
        setattr(p, name + "_w", w.data_ptr())
        setattr(p, name + "_b", b.data_ptr())
        keep += [w, b]
    return p, keep


def mlp_pack(params, out=None, precisions=None):
    """nb_mlp_pack(_sections): re-order the decoder weights into MFMA fragment order -> 1-D fp32 blob.  `precisions`:
    iterable of arithmetic names whose sections are needed (None: all)."""
    p, keep = _mlp_params(params)
    dev = keep[0].device
    if out is None:
        out = torch.zeros(mlp_pack_size(), dtype=torch.float32, device=dev)  # sections not asked for stay zero, never stale
    _req(out, torch.float32, (mlp_pack_size(),), "packed")
    bits = 3 if precisions is None else (1 | sum(_lib.PACK_SECTIONS[q] for q in set(precisions)))
    check(_lib.lib().nb_mlp_pack_sections(C.byref(p), ptr(out), int(bits), _stream()), "nb_mlp_pack_sections")
    return out


def six_bit_small_fraction(packed):
    """Per layer the 'f16f6' kernel runs with six-bit cross terms (fc_1, fc_2, the folded feature_fc / latent_fc / view_fc
    layer): the share of non-zero weights below 1/8 of their (row, 32 K) block's maximum, counted while that section was
    packed (nb_mlp_six_bit_stats_offset).  Device tensor [3]."""
    off = int(_lib.lib().nb_mlp_six_bit_stats_offset())
    c = packed[off:off + 6].view(torch.int32).reshape(3, 2).to(torch.float32)
    return c[:, 0] / c[:, 1].clamp_min(1.0)


def mlp_latent_bias(params, latent_row, out=None):
    p, keep = _mlp_params(params)
    latent_row = latent_row.detach()
    _req(latent_row, torch.float32, (128,), "latent_row")
    n = int(_lib.lib().nb_mlp_latent_bias_size())
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=latent_row.device)
    _req(out, torch.float32, (n,), "latent_bias")
    check(_lib.lib().nb_mlp_latent_bias(C.byref(p), ptr(latent_row), ptr(out), _stream()), "nb_mlp_latent_bias")
    return out


def decode_points(scene, packed, latent_bias, wpts, viewdir=None, density_only=False, debug=False, precision="f32"):
    """nb_decode_points: wpts [n,3] (+ viewdir [n,3]) -> raw [n,4] (or sigma [n,1])."""
    sc, _keep = scene
    _req(packed, torch.float32, (mlp_pack_size(),), "packed")
    _req(wpts, torch.float32, (None, 3), "wpts")
    n = wpts.shape[0]
    if not density_only:
        _req(viewdir, torch.float32, (n, 3), "viewdir")
        _req(latent_bias, torch.float32, (int(_lib.lib().nb_mlp_latent_bias_size()),), "latent_bias")
    out = torch.empty((n, 1 if density_only else 4), dtype=torch.float32, device=wpts.device)
    dbg = None
    if debug and density_only:
        raise ValueError("decode_points: the activation tap (debug) is written by the colour decode only; density_only has none")
    if debug:  # the kernel writes columns [0, 1594) of every point's tap; the six pad columns are cleared (a strided 1.5 MB fill,
        dbg = torch.empty((n, DBG_WIDTH), dtype=torch.float32, device=wpts.device)  # not the 419 MB of a training batch)
        dbg[:, TAP["PE"][1]:].zero_()
    check(_lib.lib().nb_decode_points(C.byref(sc), ptr(packed), ptr(latent_bias), ptr(wpts), ptr(viewdir), n,
                                      1 if density_only else 0, ptr(out), ptr(dbg), _lib.PRECISIONS[precision],
                                      _stream()), "nb_decode_points")
    return (out, dbg) if debug else out


def march(scene, packed, latent_bias, ray_o, ray_d, near, far, t_vals, t_rand=None, white_bkgd=False,
          want_raw=False, precision="f32", ray_order=None, cull=None, order_covers_all=False, fixup=True):
    """nb_march: all rays of one batch element -> dict of per-ray outputs.  With a `ray_order` the kernel stores only the rays
    its slots name: unless the caller vouches that every ray has a slot (`order_covers_all`, e.g. the slot list of a fully
    covered image) the outputs start out as zeros, so a ray without a slot reads 0 and never uninitialised memory.
    `fixup` (precision 'f16f6'): hand the call the scratch of its last-sample fix-up (include/nb_hip.h, `ill_scratch`) — True:
    room for min(n, FIXUP_CAP) rays, an int: for that many; the result then carries it as 'ill_scratch' (int32 view: [rays
    listed, rays whose 1e10-interval step changed side, ...])."""
    sc, _keep = scene
    _req(packed, torch.float32, (mlp_pack_size(),), "packed")
    _req(latent_bias, torch.float32, (int(_lib.lib().nb_mlp_latent_bias_size()),), "latent_bias")
    _req(ray_o, torch.float32, (None, 3), "ray_o")
    n = ray_o.shape[0]
    _req(ray_d, torch.float32, (n, 3), "ray_d")
    _req(near, torch.float32, (n,), "near")
    _req(far, torch.float32, (n,), "far")
    _req(t_vals, torch.float32, (None,), "t_vals")
    S = t_vals.shape[0]
    if t_rand is not None:
        _req(t_rand, torch.float32, (n, S), "t_rand")
    n_slots = 0
    if ray_order is not None:
        _req(ray_order, torch.int32, (None,), "ray_order")
        n_slots = int(ray_order.shape[0])
        if n_slots == 0 or n_slots % 64:
            raise ValueError("ray_order holds %d slots: a positive multiple of 64 (ops.tile_slots)" % n_slots)
    dev = ray_o.device
    alloc = torch.zeros if (ray_order is not None and not order_covers_all) else torch.empty
    rgb = alloc((n, 3), dtype=torch.float32, device=dev)
    disp = alloc((n,), dtype=torch.float32, device=dev)
    acc = alloc((n,), dtype=torch.float32, device=dev)
    weights = alloc((n, S), dtype=torch.float32, device=dev)
    depth = alloc((n,), dtype=torch.float32, device=dev)
    raw = alloc((n, S, 4), dtype=torch.float32, device=dev) if want_raw else None
    ill, ill_bytes = None, 0
    if fixup and precision == "f16f6" and n > 0:
        ill_bytes = _lib.ill_scratch_bytes(min(n, FIXUP_CAP) if fixup is True else max(int(fixup), 1))
        ill = torch.empty(ill_bytes // 4, dtype=torch.int32, device=dev)
    ev = None
    if MARCH_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    check(_lib.lib().nb_march(C.byref(sc), ptr(packed), ptr(latent_bias), ptr(ray_o), ptr(ray_d), ptr(near), ptr(far),
                              n, S, ptr(t_vals), ptr(t_rand), ptr(ray_order), n_slots,
                              C.byref(cull[0]) if cull is not None else None, 1 if white_bkgd else 0, ptr(rgb), ptr(disp),
                              ptr(acc),
                              ptr(weights), ptr(depth), ptr(raw), ptr(ill), ill_bytes, _lib.PRECISIONS[precision], _stream()), "nb_march")
    if ev is not None:
        ev[1].record()
        MARCH_EVENTS.append(ev)
    ret = {"rgb_map": rgb, "disp_map": disp, "acc_map": acc, "weights": weights, "depth_map": depth}
    if want_raw:
        ret["raw"] = raw
    if ill is not None:
        ret["ill_scratch"] = ill
    return ret


def composite(raw, z_vals, ray_d, white_bkgd=False):
    """nb_composite: raw2outputs on device. raw [n,S,4], z_vals [n,S], ray_d [n,3]."""
    _req(raw, torch.float32, (None, None, 4), "raw")
    n, S = raw.shape[:2]
    _req(z_vals, torch.float32, (n, S), "z_vals")
    _req(ray_d, torch.float32, (n, 3), "ray_d")
    dev = raw.device
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((n,), dtype=torch.float32, device=dev)
    acc = torch.empty((n,), dtype=torch.float32, device=dev)
    weights = torch.empty((n, S), dtype=torch.float32, device=dev)
    depth = torch.empty((n,), dtype=torch.float32, device=dev)
    check(_lib.lib().nb_composite(ptr(raw), ptr(z_vals), ptr(ray_d), n, S, 1 if white_bkgd else 0, ptr(rgb), ptr(disp),
                                  ptr(acc), ptr(weights), ptr(depth), _stream()), "nb_composite")
    return rgb, disp, acc, weights, depth


# --------------------------------------------------------------------------------- encoder
def scan_scratch(n, device):
    return torch.empty(int(_lib.lib().nb_scan_scratch_size(int(n))), dtype=torch.uint8, device=device)


def _i3(v):
    return (C.c_int32 * 3)(int(v[0]), int(v[1]), int(v[2]))


def enc_voxelize(coord, dhw, buf=None, grid=None):
    """nb_enc_voxelize: coord [n,3] int32 (d,h,w) -> (grid [D,H,W] i32, rows_vert, rows_lin, n_rows[1]).
    buf: a ZEROED int32 [2 * max(n, 1) + 1] buffer of the caller's for the three outputs (the encoder clears the index buffers of
    all its levels with one fill); grid: an int32 [D,H,W] buffer ALREADY FILLED WITH -1 (likewise one fill for all levels)."""
    _req(coord, torch.int32, (None, 3), "coord")
    n = coord.shape[0]
    dev = coord.device
    prefilled = grid is not None
    if grid is None:
        grid = torch.empty([int(s) for s in dhw], dtype=torch.int32, device=dev)
    _req(grid, torch.int32, tuple(int(s) for s in dhw), "grid")
    m = max(n, 1)
    if buf is None:
        buf = torch.zeros(2 * m + 1, dtype=torch.int32, device=dev)  # one fill
    else:
        _req(buf, torch.int32, (2 * m + 1,), "buf")
    rows_vert, rows_lin, n_rows = buf[:m], buf[m:2 * m], buf[2 * m:]
    scratch = scan_scratch(n, dev)
    check(_lib.lib().nb_enc_voxelize(ptr(coord), n, _i3(dhw), ptr(grid), ptr(rows_vert), ptr(rows_lin), ptr(n_rows),
                                     ptr(scratch), 1 if prefilled else 0, _stream()), "nb_enc_voxelize")
    return grid, rows_vert, rows_lin, n_rows


def down_dhw(dhw):
    return [(int(s) - 1) // 2 + 1 for s in dhw]


def down_capacity(n_in_max, in_dhw):
    """Row capacity of the strided convolution's output index set: every input row has at most 8 parents."""
    out_dhw = down_dhw(in_dhw)
    return max(min(8 * int(n_in_max), out_dhw[0] * out_dhw[1] * out_dhw[2]), 1)


def enc_downsample_index(in_lin, n_in, n_in_max, in_dhw, scratch=None, buf=None, grid=None):
    """nb_enc_downsample_index -> (out_grid, out_lin, n_out[1], n_out_max, out_dhw).
    buf: a ZEROED int32 [down_capacity(n_in_max, in_dhw) + 1] buffer of the caller's for out_lin and n_out."""
    _req(in_lin, torch.int32, (None,), "in_lin")
    _req(n_in, torch.int32, (1,), "n_in")
    dev = in_lin.device
    out_dhw = down_dhw(in_dhw)
    nvox = out_dhw[0] * out_dhw[1] * out_dhw[2]
    n_out_max = down_capacity(n_in_max, in_dhw)
    prefilled = grid is not None  # an int32 [Do,Ho,Wo] buffer already filled with -1
    out_grid = grid if prefilled else torch.empty(out_dhw, dtype=torch.int32, device=dev)
    _req(out_grid, torch.int32, tuple(out_dhw), "out_grid")
    if buf is None:
        buf = torch.zeros(n_out_max + 1, dtype=torch.int32, device=dev)  # one fill
    else:
        _req(buf, torch.int32, (n_out_max + 1,), "buf")
    out_lin, n_out = buf[:n_out_max], buf[n_out_max:]
    if scratch is None:
        scratch = scan_scratch(nvox, dev)
    check(_lib.lib().nb_enc_downsample_index(ptr(in_lin), ptr(n_in), int(n_in_max), _i3(in_dhw), _i3(out_dhw),
                                             ptr(out_grid), ptr(out_lin), ptr(n_out), n_out_max, ptr(scratch),
                                             1 if prefilled else 0, _stream()), "nb_enc_downsample_index")
    return out_grid, out_lin, n_out, n_out_max, out_dhw


def enc_downsample_index_all(in_lin, n_in, n_in_max, in_dhw, bufs, grids, scratch=None):
    """nb_enc_downsample_index_all: the index sets of len(bufs) successive strided levels in three launches -> a list of
    (out_grid, out_lin, n_out[1], n_out_max, out_dhw), one per level, equal to chained enc_downsample_index calls.
    bufs[l]: a ZEROED int32 [capacity_l + 1] buffer for out_lin and n_out (capacity_l = down_capacity of the level above);
    grids[l]: an int32 [D_l, H_l, W_l] buffer ALREADY FILLED WITH -1."""
    _req(in_lin, torch.int32, (None,), "in_lin")
    _req(n_in, torch.int32, (1,), "n_in")
    n_levels = len(bufs)
    if n_levels != len(grids) or not 1 <= n_levels <= 4:
        raise ValueError("1..4 levels, one buffer and one grid each")
    dev = in_lin.device
    out, cap, d = [], int(n_in_max), [int(x) for x in in_dhw]
    for l in range(n_levels):
        cap, d = down_capacity(cap, d), down_dhw(d)
        _req(grids[l], torch.int32, tuple(d), "grids[%d]" % l)
        _req(bufs[l], torch.int32, (cap + 1,), "bufs[%d]" % l)
        out.append((grids[l], bufs[l][:cap], bufs[l][cap:], cap, list(d)))
    if scratch is None:
        scratch = scan_scratch(max(math.prod(out[0][4]), 64), dev)
    pg = (C.c_void_p * n_levels)(*[o[0].data_ptr() for o in out])
    pl = (C.c_void_p * n_levels)(*[o[1].data_ptr() for o in out])
    pn = (C.c_void_p * n_levels)(*[o[2].data_ptr() for o in out])
    caps = (C.c_int32 * n_levels)(*[o[3] for o in out])
    check(_lib.lib().nb_enc_downsample_index_all(ptr(in_lin), ptr(n_in), int(n_in_max), _i3(in_dhw), n_levels, pg, pl, pn, caps,
                                                 ptr(scratch), 1, _stream()), "nb_enc_downsample_index_all")
    return out


def enc_conv(in_rows, in_grid, in_dhw, out_lin, n_out, n_out_max, out_dhw, stride, weight, stats=None):
    """nb_enc_conv -> (out_rows [n_out_max, Cout], stats [2*Cout] fp64).  `stats`: a ZEROED fp64 [2*Cout] buffer of the
    caller's (the encoder clears the statistics of all its layers with one fill) — else the call allocates and clears one."""
    _req(weight, torch.float32, (3, 3, 3, None, None), "conv weight")
    cin, cout = int(weight.shape[3]), int(weight.shape[4])
    _req(in_rows, torch.float32, (None, cin), "in_rows")
    _req(in_grid, torch.int32, tuple(int(s) for s in in_dhw), "in_grid")
    _req(out_lin, torch.int32, (None,), "out_lin")
    _req(n_out, torch.int32, (1,), "n_out")
    if out_lin.shape[0] < n_out_max:
        raise ValueError("out_lin shorter than n_out_max")
    dev = in_rows.device
    out_rows = torch.empty((max(int(n_out_max), 1), cout), dtype=torch.float32, device=dev)
    flags = 0 if stats is None else 1  # NB_CONV_STATS_ZEROED
    if stats is None:
        stats = torch.empty(2 * cout, dtype=torch.float64, device=dev)
    else:
        _req(stats, torch.float64, (2 * cout,), "stats")
    check(_lib.lib().nb_enc_conv(ptr(in_rows), ptr(in_grid), _i3(in_dhw), ptr(out_lin), ptr(n_out), int(n_out_max),
                                 _i3(out_dhw), int(stride), ptr(weight), cin, cout, ptr(out_rows), ptr(stats), flags,
                                 _stream()), "nb_enc_conv")
    return out_rows, stats


def enc_bn_relu(rows, n_rows, n_rows_max, stats, gamma, beta, running_mean, running_var, training, eps,
                rows_lin=None, dense=None, momentum=-1.0, rows_out=None):
    """nb_enc_bn_relu (in place on rows) -> batch_stats [2C+1] = mean | biased var | n_rows.
    momentum >= 0 (training only): running_mean / running_var are updated in place by the kernel."""
    c = int(rows.shape[1])
    _req(rows, torch.float32, (None, c), "rows")
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        _req(t, torch.float32, (c,), nm)
    if stats is not None:
        _req(stats, torch.float64, (2 * c,), "stats")
    if dense is not None:
        _req(dense, torch.float32, (None, None, None, c), "dense")
        _req(rows_lin, torch.int32, (None,), "rows_lin")
    if rows_out is not None:
        _req(rows_out, torch.float32, tuple(rows.shape), "rows_out")
    batch_stats = torch.empty(2 * c + 1, dtype=torch.float32, device=rows.device)
    check(_lib.lib().nb_enc_bn_relu(ptr(rows), ptr(n_rows), int(n_rows_max), c, ptr(stats), ptr(gamma), ptr(beta),
                                    ptr(running_mean), ptr(running_var), 1 if training else 0, float(eps),
                                    float(momentum), ptr(batch_stats), ptr(rows_lin), ptr(dense), ptr(rows_out),
                                    _stream()), "nb_enc_bn_relu")
    return batch_stats


def enc_conv_pack16(weight, backward_input=False):
    """nb_enc_conv_pack16: spconv-layout fp32 weight -> head / remainder B fragments (int16 tensor): fp16 pairs of the forward
    convolution, or (backward_input) bf16 pairs of the stride-1 layer's backward-input convolution (mirrored offsets,
    transposed slabs: a convolution with Cout input and Cin output channels)."""
    _req(weight, torch.float32, (3, 3, 3, None, None), "conv weight")
    cin, cout = int(weight.shape[3]), int(weight.shape[4])
    packed = torch.empty(27 * cin * cout * 2, dtype=torch.int16, device=weight.device)
    if backward_input:
        check(_lib.lib().nb_enc_conv_pack16(ptr(weight), cout, cin, ptr(packed), 1, _stream()), "nb_enc_conv_pack16")
    else:
        check(_lib.lib().nb_enc_conv_pack16(ptr(weight), cin, cout, ptr(packed), 0, _stream()), "nb_enc_conv_pack16")
    return packed


def enc_conv_pack16_batch(jobs):
    """nb_enc_conv_pack16_batch: jobs = [(weight [3,3,3,Cin,Cout], backward_input)] -> list of packed int16 tensors, one launch
    (per 32 jobs)."""
    outs = []
    for i0 in range(0, len(jobs), 32):
        chunk = jobs[i0:i0 + 32]
        n = len(chunk)
        w_p, out_p = (C.c_void_p * n)(), (C.c_void_p * n)()
        cin_a, cout_a, mode_a = (C.c_int32 * n)(), (C.c_int32 * n)(), (C.c_int32 * n)()
        for i, (weight, bwd) in enumerate(chunk):
            _req(weight, torch.float32, (3, 3, 3, None, None), "conv weight")
            ci, co = int(weight.shape[3]), int(weight.shape[4])
            packed = torch.empty(27 * ci * co * 2, dtype=torch.int16, device=weight.device)
            outs.append(packed)
            w_p[i], out_p[i] = weight.data_ptr(), packed.data_ptr()
            cin_a[i], cout_a[i], mode_a[i] = (co, ci, 1) if bwd else (ci, co, 0)
        check(_lib.lib().nb_enc_conv_pack16_batch(n, w_p, cin_a, cout_a, out_p, mode_a, _stream()), "nb_enc_conv_pack16_batch")
    return outs


def enc_conv16(in_split, in_grid, in_dhw, out_lin, n_out, n_out_max, out_dhw, stride, wpacked, cin, cout, stats=None,
               bf16=False):
    """nb_enc_conv16 on split rows (int16 [2, cap, Cin]: fp16 heads | remainders) -> (out_rows fp32, stats fp64); `stats` as
    in enc_conv.  bf16: the planes and the packed weight are bf16 pairs (the backward-input convolution)."""
    _req(in_split, torch.int16, (2, None, cin), "in_split")
    _req(wpacked, torch.int16, (27 * cin * cout * 2,), "wpacked")
    _req(in_grid, torch.int32, tuple(int(s) for s in in_dhw), "in_grid")
    _req(out_lin, torch.int32, (None,), "out_lin")
    _req(n_out, torch.int32, (1,), "n_out")
    if out_lin.shape[0] < n_out_max:
        raise ValueError("out_lin shorter than n_out_max")
    dev = in_split.device
    out_rows = torch.empty((max(int(n_out_max), 1), cout), dtype=torch.float32, device=dev)
    flags = (0 if stats is None else 1) | (2 if bf16 else 0)  # NB_CONV_STATS_ZEROED | NB_CONV_BF16
    if stats is None:
        stats = torch.empty(2 * cout, dtype=torch.float64, device=dev)
    else:
        _req(stats, torch.float64, (2 * cout,), "stats")
    check(_lib.lib().nb_enc_conv16(ptr(in_split), int(in_split.shape[1]), ptr(in_grid), _i3(in_dhw), ptr(out_lin), ptr(n_out),
                                   int(n_out_max), _i3(out_dhw), int(stride), ptr(wpacked), cin, cout, ptr(out_rows),
                                   ptr(stats), flags, _stream()), "nb_enc_conv16")
    return out_rows, stats


def enc_bn_relu_split(rows, n_rows, n_rows_max, stats, gamma, beta, running_mean, running_var, training, eps,
                      rows_lin=None, dense=None, momentum=-1.0, rows_out=None):
    """nb_enc_bn_relu_split -> (rows_split int16 [2, n_rows_max, C], batch_stats); rows_out (fp32, same shape as rows):
    receives the activated rows in fp32 as well (training forward)."""
    c = int(rows.shape[1])
    _req(rows, torch.float32, (None, c), "rows")
    if rows_out is not None:
        _req(rows_out, torch.float32, tuple(rows.shape), "rows_out")
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        _req(t, torch.float32, (c,), nm)
    if stats is not None:
        _req(stats, torch.float64, (2 * c,), "stats")
    if dense is not None:
        _req(dense, torch.float32, (None, None, None, c), "dense")
        _req(rows_lin, torch.int32, (None,), "rows_lin")
    n_rows_max = max(int(n_rows_max), 1)
    if rows.shape[0] < n_rows_max:
        raise ValueError("rows shorter than n_rows_max")
    split = torch.empty((2, n_rows_max, c), dtype=torch.int16, device=rows.device)
    batch_stats = torch.empty(2 * c + 1, dtype=torch.float32, device=rows.device)
    check(_lib.lib().nb_enc_bn_relu_split(ptr(rows), ptr(n_rows), n_rows_max, c, ptr(stats), ptr(gamma), ptr(beta),
                                          ptr(running_mean), ptr(running_var), 1 if training else 0, float(eps),
                                          float(momentum), ptr(batch_stats), ptr(rows_lin), ptr(dense), ptr(split),
                                          ptr(rows_out), _stream()), "nb_enc_bn_relu_split")
    return split, batch_stats


def enc_gather_codes(codes, rows_vert, n_rows, n_rows_max):
    _req(codes, torch.float32, (None, None), "codes")
    _req(rows_vert, torch.int32, (None,), "rows_vert")
    c = int(codes.shape[1])
    rows = torch.empty((max(int(n_rows_max), 1), c), dtype=torch.float32, device=codes.device)
    check(_lib.lib().nb_enc_gather_codes(ptr(codes), ptr(rows_vert), ptr(n_rows), int(n_rows_max), c, ptr(rows),
                                         _stream()), "nb_enc_gather_codes")
    return rows


# --------------------------------------------------------------------------------- ray generation
def raygen(H, W, K, R, T, bounds, device):
    """nb_raygen: K, R [3,3], T [3] host float64 arrays; bounds [2,3] host float32 (world AABB).
    Returns device tensors (ray_o, ray_d, near, far, mask_at_box, n_rays[1]) with H*W capacity; the
    first n_rays rows are valid."""
    import numpy as np

    K = np.asarray(K, np.float64).reshape(9)
    R = np.asarray(R, np.float64).reshape(9)
    T = np.asarray(T, np.float64).reshape(3)
    b = np.asarray(bounds, np.float32).reshape(6)
    n = int(H) * int(W)
    ray_o = torch.empty((n, 3), dtype=torch.float32, device=device)
    ray_d = torch.empty((n, 3), dtype=torch.float32, device=device)
    near = torch.empty(n, dtype=torch.float32, device=device)
    far = torch.empty(n, dtype=torch.float32, device=device)
    mask = torch.empty(n, dtype=torch.uint8, device=device)
    n_rays = torch.zeros(1, dtype=torch.int32, device=device)
    scratch = scan_scratch(n, device)
    with torch.cuda.device(device):
        check(_lib.lib().nb_raygen(int(H), int(W), (C.c_double * 9)(*K), (C.c_double * 9)(*R), (C.c_double * 3)(*T),
                                   (C.c_float * 6)(*b), ptr(ray_o), ptr(ray_d), ptr(near), ptr(far), ptr(mask),
                                   ptr(n_rays), ptr(scratch), _stream()), "nb_raygen")
    return ray_o, ray_d, near, far, mask, n_rays


def image_assemble(mask_at_box, rgb_map, depth_map=None, white_bkgd=False, bgr=False, scale=1.0):
    """nb_image_assemble: mask_at_box [H*W] uint8/bool (any shape, flattened), rgb_map [n,3] and optional depth_map [n]
    in compacted pixel order -> (img [H*W,3], depth [H*W] or None) on device."""
    mask = mask_at_box.reshape(-1)
    if mask.dtype == torch.bool:
        mask = mask.to(torch.uint8)
    _req(mask, torch.uint8, (None,), "mask_at_box")
    _req(rgb_map, torch.float32, (None, 3), "rgb_map")
    n_pix, n = mask.shape[0], rgb_map.shape[0]
    if depth_map is not None:
        _req(depth_map, torch.float32, (n,), "depth_map")
    img = torch.empty((n_pix, 3), dtype=torch.float32, device=mask.device)
    depth = torch.empty(n_pix, dtype=torch.float32, device=mask.device) if depth_map is not None else None
    scratch = scan_scratch(n_pix, mask.device)
    with torch.cuda.device(mask.device):
        check(_lib.lib().nb_image_assemble(ptr(mask), n_pix, ptr(rgb_map), ptr(depth_map), n, 1 if white_bkgd else 0,
                                           1 if bgr else 0, float(scale), ptr(img), ptr(depth), ptr(scratch), _stream()),
              "nb_image_assemble")
    return img, depth


TILE = 8  # rays are marched in 8 x 8 pixel tiles (64 slots = one workgroup of the fused march), made of four 4 x 4 blocks


def tile_pixels(H, W, device):
    """Static part of the slot list of an H x W image: slot -> pixel id (row-major), -1 where the tile sticks out of the image.
    Slot s = 64 * tile + 16 * block + 4 * (y % 4) + (x % 4), tiles row-major, blocks row-major inside a tile: 64 consecutive
    slots = one compact 8 x 8 tile (the voxels its samples touch at a depth step are a few small boxes), every 16 = one 4 x 4
    block (one wave's samples), every 32 = an 8 x 4 half tile."""
    ht, wt = (H + TILE - 1) // TILE, (W + TILE - 1) // TILE
    s = torch.arange(ht * wt * 64, device=device)
    tile, inner = torch.div(s, 64, rounding_mode="floor"), s % 64
    blk, r = torch.div(inner, 16, rounding_mode="floor"), inner % 16
    py = torch.div(tile, wt, rounding_mode="floor") * TILE + torch.div(blk, 2, rounding_mode="floor") * 4 + torch.div(r, 4, rounding_mode="floor")
    px = (tile % wt) * TILE + (blk % 2) * 4 + r % 4
    pix = py * W + px
    return torch.where((py < H) & (px < W), pix, torch.full_like(pix, -1))


def tile_slots(mask, slot_pixels, begin=0, end=None):
    """The `ray_order` of nb_march (include/nb_hip.h) for the rays of an image: `mask` [H*W] bool/uint8 = which pixels have a
    ray (the rays are the mask's non-zeros in pixel order, batch['mask_at_box']), `slot_pixels` = tile_pixels(H, W), [begin, end)
    = the range of rays to march (a rank's share).  Every tile keeps its own 64 slots: missing pixels become padding slots
    (they march the tile's first ray and store nothing), tiles without a ray are dead.  No host synchronisation: cumulative
    sum + gathers on the device.  int32 [tiles * 64]; entries index the rays of the range (ray - begin)."""
    m = mask.reshape(-1) != 0
    idx = torch.cumsum(m.to(torch.int32), 0, dtype=torch.int32) - 1  # ray index of every pixel that has one
    ok = m & (idx >= begin)
    if end is not None:
        ok = ok & (idx < end)
    p = slot_pixels.clamp_min(0)
    ok_s = ok[p] & (slot_pixels >= 0)
    ray = idx[p] - begin
    big = 1 << 30
    first = torch.where(ok_s, ray, torch.full_like(ray, big)).view(-1, 64).amin(1, keepdim=True)
    fill = torch.where(first < big, -(first + 1), torch.full_like(first, _lib.SLOT_DEAD))
    return torch.where(ok_s.view(-1, 64), ray.view(-1, 64), fill).reshape(-1).to(torch.int32).contiguous()


# --------------------------------------------------------------------------------- backward pass
def composite_bwd(raw, z_vals, ray_d, d_rgb_map, white_bkgd=False, d_acc_map=None, d_depth_map=None):
    """nb_composite_bwd: d raw [n,S,4] from d rgb_map [n,3] (+ optional d acc_map / d depth_map [n])."""
    _req(raw, torch.float32, (None, None, 4), "raw")
    n, S = raw.shape[:2]
    _req(z_vals, torch.float32, (n, S), "z_vals")
    _req(ray_d, torch.float32, (n, 3), "ray_d")
    _req(d_rgb_map, torch.float32, (n, 3), "d_rgb_map")
    for t, nm in ((d_acc_map, "d_acc_map"), (d_depth_map, "d_depth_map")):
        if t is not None:
            _req(t, torch.float32, (n,), nm)
    d_raw = torch.empty_like(raw)
    check(_lib.lib().nb_composite_bwd(ptr(raw), ptr(z_vals), ptr(ray_d), n, S, 1 if white_bkgd else 0, ptr(d_rgb_map),
                                      ptr(d_acc_map), ptr(d_depth_map), ptr(d_raw), _stream()), "nb_composite_bwd")
    return d_raw


def _mat(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or \
            (t.shape[1] > 1 and t.stride(1) != 1) or t.stride(0) < t.shape[1]:
        raise ValueError("%s must be a 2-D fp32 HIP tensor with unit column stride" % name)
    return t


def sgemm(a, b, trans_a=False, trans_b=False, out=None, alpha=1.0, beta=0.0, relu_mask=None, colsum=None):
    """nb_gemm_fused (fp32 MFMA kernels of libnb_hip.so): out[m,n] = alpha * op(a) @ op(b) + beta * out, row-major; a / b /
    out may be column slices of wider matrices (row stride = leading dimension).  Epilogues of the backward chain (not with
    trans_a): relu_mask [m, n] (slice allowed): out is zeroed where relu_mask <= 0; colsum [n]: += column sums of out."""
    _mat(a, "a")
    _mat(b, "b")
    m, k = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    k2, n = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if k != k2:
        raise ValueError("sgemm: inner dimensions %d vs %d" % (k, k2))
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
        beta = 0.0
    _mat(out, "out")
    if tuple(out.shape) != (m, n):
        raise ValueError("sgemm: out is %s, expected %s" % (tuple(out.shape), (m, n)))
    ldy = 0
    if relu_mask is not None:
        _mat(relu_mask, "relu_mask")
        if tuple(relu_mask.shape) != (m, n):
            raise ValueError("sgemm: relu_mask is %s, expected %s" % (tuple(relu_mask.shape), (m, n)))
        ldy = relu_mask.stride(0)
    if colsum is not None:
        _req(colsum, torch.float32, (n,), "colsum")
    if (relu_mask is not None or colsum is not None) and trans_a:
        raise ValueError("sgemm: epilogues are not available for the weight-gradient form (trans_a)")
    check(_lib.lib().nb_gemm_fused(1 if trans_a else 0, 1 if trans_b else 0, m, n, k, float(alpha), ptr(a), a.stride(0), ptr(b),
                                   b.stride(0), float(beta), ptr(out), out.stride(0), ptr(relu_mask), ldy, ptr(colsum),
                                   _stream()), "nb_gemm_fused")
    return out


def relu_bwd_(dy, y):
    """nb_relu_bwd in place: dy *= (y > 0); y is the post-activation value."""
    _req(dy, torch.float32, None, "dy")
    _req(y, torch.float32, tuple(dy.shape), "y")
    check(_lib.lib().nb_relu_bwd(ptr(dy), ptr(y), dy.numel(), _stream()), "nb_relu_bwd")
    return dy


def colsum(x, out=None):
    """nb_colsum: out[c] += sum_r x[r, c] (x may be a column slice)."""
    _mat(x, "x")
    if out is None:
        out = torch.zeros(x.shape[1], dtype=torch.float32, device=x.device)
    _req(out, torch.float32, (x.shape[1],), "out")
    check(_lib.lib().nb_colsum(ptr(x), x.shape[0], x.shape[1], x.stride(0), ptr(out), _stream()), "nb_colsum")
    return out


def trilinear_bwd(scene, grids, drows, wpts, d_feat, run_length=1):
    """nb_trilinear_bwd: d_feat [n,352] -> drows[l] += gradient of the active rows of level l.  run_length: the points come as
    runs of that many consecutive samples of one ray (contributions to one voxel are pre-summed along a run)."""
    sc, _keep = scene
    _req(wpts, torch.float32, (None, 3), "wpts")
    n = wpts.shape[0]
    _req(d_feat, torch.float32, (n, 352), "d_feat")
    g4 = (C.c_void_p * 4)()
    d4 = (C.c_void_p * 4)()
    for l in range(4):
        _req(grids[l], torch.int32, tuple(int(v) for v in sc.vol_dhw[l]), "grid[%d]" % l)
        _req(drows[l], torch.float32, (None, LEVEL_CHANNELS[l]), "drows[%d]" % l)
        g4[l] = grids[l].data_ptr()
        d4[l] = drows[l].data_ptr()
    check(_lib.lib().nb_trilinear_bwd(C.byref(sc), g4, d4, ptr(wpts), ptr(d_feat), n, int(run_length), _stream()),
          "nb_trilinear_bwd")
    return drows


class ZeroArena:
    """One zero fill for everything a pass accumulates into (atomics, column sums, scatter targets): `take(shape, dtype)` hands
    out views of a single torch.zeros buffer sized by a planning pass over the same requests (`ZeroArena.plan()` counts)."""

    def __init__(self, n_bytes, device):
        self.buf = torch.zeros((int(n_bytes) + 7) // 8, dtype=torch.float64, device=device).view(torch.uint8)
        self.slack = self.buf.numel() - int(n_bytes)  # rounding of the buffer to whole doubles
        self.off = 0

    @staticmethod
    def size_of(requests):
        return sum((math.prod(shape) * torch.empty((), dtype=dt).element_size() + 15) // 16 * 16 for shape, dt in requests)

    def take(self, shape, dtype=torch.float32):
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        n = math.prod(shape) * torch.empty((), dtype=dtype).element_size()
        if self.off + n > self.buf.numel():
            raise RuntimeError("ZeroArena exhausted: planned %d bytes, asked for %d more at %d" % (self.buf.numel(), n, self.off))
        t = self.buf[self.off:self.off + n].view(dtype).view(shape)
        self.off += (n + 15) // 16 * 16
        return t


def enc_bn_relu_bwd(dy, y, x, n_rows, n_rows_max, batch_stats, eps, gamma, want_split=False, sums=None):
    """nb_enc_bn_relu_bwd -> (dx, dgamma, dbeta) or, with want_split, (dx, dgamma, dbeta, dx_split int16 [2, rows, C]: dx as bf16
    head / remainder planes for the backward-input convolution on the matrix pipe)."""
    c = int(x.shape[1])
    for t, nm in ((dy, "dy"), (y, "y"), (x, "x")):
        _req(t, torch.float32, (None, c), nm)
    _req(batch_stats, torch.float32, (2 * c + 1,), "batch_stats")
    _req(gamma, torch.float32, (c,), "gamma")
    dev = x.device
    zeroed = sums is not None  # a ZeroArena view: the call's own memset is skipped
    if sums is None:
        sums = torch.empty(2 * c, dtype=torch.float64, device=dev)
    _req(sums, torch.float64, (2 * c,), "sums")
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    split = torch.empty((2, max(int(n_rows_max), 1), c), dtype=torch.int16, device=dev) if want_split else None
    check(_lib.lib().nb_enc_bn_relu_bwd(ptr(dy), ptr(y), ptr(x), ptr(n_rows), int(n_rows_max), c, ptr(batch_stats),
                                        float(eps), ptr(gamma), ptr(sums), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(split),
                                        1 if zeroed else 0, _stream()), "nb_enc_bn_relu_bwd")
    return (dx, dgamma, dbeta, split) if want_split else (dx, dgamma, dbeta)


def enc_conv_bwd_input(dx, out_grid, out_dhw, in_lin, n_in, n_in_max, in_dhw, stride, weight, out=None):
    """nb_enc_conv_bwd_input -> din [n_in_max, Cin]; out: a ZEROED buffer of that shape (ZeroArena), default torch.zeros."""
    cin, cout = int(weight.shape[3]), int(weight.shape[4])
    _req(dx, torch.float32, (None, cout), "dx")
    _req(out_grid, torch.int32, tuple(int(s) for s in out_dhw), "out_grid")
    _req(in_lin, torch.int32, (None,), "in_lin")
    din = out if out is not None else torch.zeros((max(int(n_in_max), 1), cin), dtype=torch.float32, device=dx.device)
    _req(din, torch.float32, (max(int(n_in_max), 1), cin), "din")
    check(_lib.lib().nb_enc_conv_bwd_input(ptr(dx), ptr(out_grid), _i3(out_dhw), ptr(in_lin), ptr(n_in), int(n_in_max),
                                           _i3(in_dhw), int(stride), ptr(weight), cin, cout, ptr(din), _stream()),
          "nb_enc_conv_bwd_input")
    return din


def enc_conv_bwd_weight(in_rows, in_grid, in_dhw, out_lin, n_out, n_out_max, out_dhw, stride, dx, cin, cout, dx_split=None, out=None,
                        rulebook=None):
    """nb_enc_conv_bwd_weight -> dW [3,3,3,Cin,Cout].  dx_split (int16 [2, n_out_max, Cout], enc_bn_relu_bwd(want_split=True)):
    the product runs on the 16-bit matrix pipe with bf16 pairs (Cin >= 32)."""
    _req(in_rows, torch.float32, (None, cin), "in_rows")
    _req(dx, torch.float32, (None, cout), "dx")
    if dx_split is not None:
        _req(dx_split, torch.int16, (2, max(int(n_out_max), 1), cout), "dx_split")
    zeroed = out is not None  # a ZeroArena view: the call's own memset is skipped
    dw = out if zeroed else torch.empty((3, 3, 3, cin, cout), dtype=torch.float32, device=dx.device)
    _req(dw, torch.float32, (3, 3, 3, cin, cout), "dweight")
    # rulebook: a one-element list cache shared by the layers of one (input grid, output rows, stride): the first call fills it,
    # the others reuse the neighbour table (the three submanifold layers of a level have the same one)
    ready = rulebook is not None and len(rulebook) == 1
    rb = rulebook[0] if ready else torch.empty(max(int(n_out_max), 1) * 27, dtype=torch.int32, device=dx.device)
    _req(rb, torch.int32, (max(int(n_out_max), 1) * 27,), "rulebook")
    check(_lib.lib().nb_enc_conv_bwd_weight(ptr(in_rows), ptr(in_grid), _i3(in_dhw), ptr(out_lin), ptr(n_out),
                                            int(n_out_max), _i3(out_dhw), int(stride), ptr(dx), ptr(dx_split), cin, cout,
                                            ptr(dw), ptr(rb), (1 if zeroed else 0) | (2 if ready else 0), _stream()),
          "nb_enc_conv_bwd_weight")
    if rulebook is not None and not ready:
        rulebook.append(rb)
    return dw


def enc_scatter_codes_bwd(drows, rows_vert, n_rows, n_rows_max, n_codes, out=None):
    c = int(drows.shape[1])
    dcodes = out if out is not None else torch.zeros((n_codes, c), dtype=torch.float32, device=drows.device)
    _req(dcodes, torch.float32, (n_codes, c), "dcodes")
    check(_lib.lib().nb_enc_scatter_codes_bwd(ptr(drows), ptr(rows_vert), ptr(n_rows), int(n_rows_max), c, ptr(dcodes),
                                              _stream()), "nb_enc_scatter_codes_bwd")
    return dcodes


def make_cull(masks, RT, K, R0=None, Th0=None):
    """nb_cull from DEVICE tensors: masks [n_views,H,W] uint8 (non-zero = inside), RT [n_views,3,4], K [n_views,3,3]
    (batch['msks'][0], batch['RT'][0], batch['Ks'][0]); R0 [3,3] / Th0 [3] select the _msk variant (snapshot-frame
    placement).  Nothing visits the host.  Returns (NbCull, keepalive)."""
    masks = masks if masks.dtype == torch.uint8 else masks.to(torch.uint8)
    masks = masks.contiguous()
    _req(masks, torch.uint8, (None, None, None), "masks")
    nv, H, W = (int(v) for v in masks.shape)
    if not 1 <= nv <= 64:
        raise ValueError("1..64 mask views supported, got %d" % nv)
    cam = torch.cat([RT.detach().reshape(nv, 12).float(), K.detach().reshape(nv, 9).float()], 1).contiguous()
    _req(cam, torch.float32, (nv, 21), "cam")
    c = NbCull()
    c.n_views, c.H, c.W = nv, H, W
    c.pre_affine = 0 if R0 is None else 1
    c.msk = masks.data_ptr()
    c.cam = cam.data_ptr()
    keep = [masks, cam]
    if R0 is not None:
        snap = torch.cat([R0.detach().reshape(9).float(), Th0.detach().reshape(-1)[:3].float()]).contiguous()
        _req(snap, torch.float32, (12,), "snap")
        c.snap = snap.data_ptr()
        keep.append(snap)
    return c, keep
