"""`Network` — drop-in for zju3dv/neuralbody lib/networks/latent_xyzc.py::Network whose compute is
the HIP library (no spconv, no torch ops on the hot path).

Kept from the reference:
  * parameter / buffer names and shapes -> `state_dict()` is interchangeable (120 keys:
    c.weight, xyzc_net.convN.K.weight, xyzc_net.convN.K+1.{weight,bias,running_mean,running_var,
    num_batches_tracked}, latent.weight, fc_0.{weight,bias} [out,in,1], ...)      latent_xyzc.py:10-28,166-182
  * encode_sparse_voxels(sp_input) -> list of 4 volumes [B,C,D,H,W]               latent_xyzc.py:30-39
  * calculate_density(wpts, feature_volume, sp_input) -> [B,N,1]                  latent_xyzc.py:74-89
  * calculate_density_color(wpts, viewdir, feature_volume, sp_input) -> [B,N,4]   latent_xyzc.py:91-126
  * forward(sp_input, grid_coords, viewdir, light_pts) -> [B,N,4]                 latent_xyzc.py:128-163
  * `.training` honoured: BatchNorm uses batch statistics over the active voxels and updates the
    running statistics (momentum 0.01) in train() — which is also how run.py renders (run.py:57,89).
New: `render_rays(...)`, the fused march used by `Renderer.render` (one launch for all rays).
"""
import math
import os

import torch
import torch.nn as nn

from . import ops

N_VERTS = 6890
CODE_DIM = 16
# (name, cin, cout, n_convs, stride) — latent_xyzc.py:170-182
ENCODER_BLOCKS = [
    ("conv0", 16, 16, 2, 1), ("down0", 16, 32, 1, 2), ("conv1", 32, 32, 2, 1), ("down1", 32, 64, 1, 2),
    ("conv2", 64, 64, 3, 1), ("down2", 64, 128, 1, 2), ("conv3", 128, 128, 3, 1), ("down3", 128, 128, 1, 2),
    ("conv4", 128, 128, 3, 1),
]
DENSE_AFTER = ("conv1", "conv2", "conv3", "conv4")  # latent_xyzc.py:188-201
BN_EPS, BN_MOMENTUM = 1e-3, 0.01  # latent_xyzc.py:215
DEFAULT_PRECISION = "auto"
ENC_SPLIT = os.environ.get("NB_ENC_SPLIT", "1") != "0"  # encoder convolutions with >= 32 input channels on the 16-bit matrix pipe
SIX_BIT_MAX_SMALL = 0.5  # precision 'auto': largest per-layer share of weights six-bit blocks cannot hold before it takes 'f32'


class FeatureVolumes(list):
    """The four volumes of `Network.encode_sparse_voxels` ([1,C,D,H,W] views of channels-last storage, as the reference
    returns them) together with the index structures they were scattered from: `sparse[l]` = (index grid [D,H,W] int32, linear
    voxel index of every active row, device-side row count [1], row capacity).  Precision 'f16f6' builds its fc_0-folded
    planes from them (`fold`: (fc_0 weight key, ops.fold_build result), rebuilt when the weight changes); a plain list of
    volumes from elsewhere gets its active set from ops.sparsify."""

    def __init__(self, volumes, sparse=None):
        super().__init__(volumes)
        self.sparse = sparse
        self.fold = None


class SparseConv3dParam(nn.Module):
    """Holds the weight of one SubMConv3d / SparseConv3d in spconv 1.x layout [kD,kH,kW,Cin,Cout]
    (bias=False).  The convolution itself runs in nb_enc_conv."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = cin, cout, stride
        bound = 1.0 / math.sqrt(27 * cin)
        self.weight = nn.Parameter(torch.empty(3, 3, 3, cin, cout).uniform_(-bound, bound))

    def extra_repr(self):
        return "%d, %d, kernel_size=3, stride=%d, bias=False" % (self.in_channels, self.out_channels, self.stride)


def _block(cin, cout, n, stride):
    layers = []
    for j in range(n):
        layers += [SparseConv3dParam(cin if j == 0 else cout, cout, stride),
                   nn.BatchNorm1d(cout, eps=BN_EPS, momentum=BN_MOMENTUM), nn.ReLU()]
    return nn.Sequential(*layers)


class SparseConvNet(nn.Module):
    """Parameter container + layer schedule of the reference's SparseConvNet (latent_xyzc.py:166-205)."""

    def __init__(self):
        super().__init__()
        for name, cin, cout, n, stride in ENCODER_BLOCKS:
            setattr(self, name, _block(cin, cout, n, stride))

    def _packed16(self, conv):
        """fp16 head / remainder B fragments of one convolution's weight, rebuilt when the parameter changes (the entry
        holds the storage it was packed from, so its address cannot be recycled under the key)."""
        w = conv.weight.detach()
        key = (w.untyped_storage(), w.data_ptr(), w._version)
        old = getattr(conv, "_nb_packed16", None)
        if old is None or old[0][0]._cdata != key[0]._cdata or old[0][1:] != key[1:]:
            old = (key, ops.enc_conv_pack16(w))
            conv._nb_packed16 = old
        return old[1]

    def forward(self, codes, coord, out_sh, training, save=None):
        """codes [6890,16] fp32, coord [6890,3] int32 (d,h,w) -> 4 channels-last volumes [D,H,W,C].
        save: optional list that receives one record per conv+BN+ReLU layer (index structures, raw and activated
        rows, batch statistics) — everything neuralbody_amd.training.encoder_backward needs."""
        dev = codes.device
        dhw = [int(s) for s in out_sh]
        n_max = coord.shape[0]
        layers = [(name, cin, cout, n, stride, j) for name, cin, cout, n, stride in ENCODER_BLOCKS for j in range(n)]
        # ONE zero fill per element type for everything the pass needs cleared: the index buffers of the levels (rows_vert |
        # rows_lin | n_rows, then out_lin | n_out per strided layer), and the dense volumes + the layers' fp64 statistics
        int_sizes, dense_shapes, cap, d = [2 * max(n_max, 1) + 1], [], n_max, dhw
        for name, cin, cout, n, stride, j in layers:
            if stride == 2:
                int_sizes.append(ops.down_capacity(cap, d) + 1)
                cap, d = int_sizes[-1] - 1, ops.down_dhw(d)
            if name in DENSE_AFTER and j == n - 1:
                dense_shapes.append(d + [cout])
        int_bufs = list(torch.zeros(sum(int_sizes), dtype=torch.int32, device=dev).split(int_sizes))
        n_stats = 2 * len(layers) * 256  # fp64 [layers, 256] in front (8-byte aligned), the volumes behind it (64-float aligned)
        dense_sizes = [(math.prod(sh) + 63) // 64 * 64 for sh in dense_shapes]
        f32_buf = torch.zeros(n_stats + sum(dense_sizes), dtype=torch.float32, device=dev)
        stats_all = f32_buf[:n_stats].view(torch.float64).view(len(layers), 256)
        dense_bufs = [b[:math.prod(sh)].view(sh) for b, sh in zip(f32_buf[n_stats:].split(dense_sizes), dense_shapes)]
        grid, rows_vert, rows_lin, n_rows = ops.enc_voxelize(coord, dhw, buf=int_bufs.pop(0))
        rows = ops.enc_gather_codes(codes, rows_vert, n_rows, n_max)
        if save is not None:
            save.append({"rows_vert": rows_vert, "n_rows": n_rows, "n_max": n_max})
        volumes = []
        sparse = []  # per volume: (index grid, rows_lin, n_rows, capacity) of the rows it was scattered from
        bn_updates = []
        # Convolutions with >= 32 input channels run on the 16-bit matrix pipe with split operands (ops.enc_conv16, three
        # products, fp32 accumulation: ~2^-21 relative per product); their input rows arrive as fp16 head / remainder planes
        # written by the producing BatchNorm kernel, which in a training forward (save given) writes the fp32 activations
        # next to them — the backward pass differentiates the exact-fp32 formulas on those.  NB_ENC_SPLIT=0 keeps every
        # layer on the exact-fp32 MFMA kernel.
        fast = ENC_SPLIT
        rows_are_split = False
        rows_f32 = rows  # the fp32 form of the current layer's input rows (what the backward record keeps)
        for li, (name, cin, cout, n, stride, j) in enumerate(layers):
            block = getattr(self, name)
            conv, bn = block[3 * j], block[3 * j + 1]
            if stride == 2:
                out_grid, out_lin, n_out, n_out_max, out_dhw = ops.enc_downsample_index(rows_lin, n_rows, n_max, dhw, buf=int_bufs.pop(0))
            else:
                out_grid, out_lin, n_out, n_out_max, out_dhw = grid, rows_lin, n_rows, n_max, dhw
            if rows_are_split:
                new_rows, stats = ops.enc_conv16(rows, grid, dhw, out_lin, n_out, n_out_max, out_dhw, stride,
                                                 self._packed16(conv), cin, cout, stats=stats_all[li, :2 * cout])
            else:
                new_rows, stats = ops.enc_conv(rows, grid, dhw, out_lin, n_out, n_out_max, out_dhw, stride,
                                               conv.weight.detach(), stats=stats_all[li, :2 * cout])
            dense = None
            if name in DENSE_AFTER and j == n - 1:
                dense = dense_bufs.pop(0)
                assert list(dense.shape) == out_dhw + [cout]
                volumes.append(dense)
                sparse.append((out_grid, out_lin, n_out, n_out_max))
            next_split = fast and li + 1 < len(layers) and cout >= 32  # the consumer of these rows is an enc_conv16
            act = torch.empty_like(new_rows) if save is not None else None  # keep the raw conv output when saving
            split = None
            if next_split:
                split, bstats = ops.enc_bn_relu_split(new_rows, n_out, n_out_max, stats, bn.weight.detach(), bn.bias.detach(),
                                                      bn.running_mean, bn.running_var, training, bn.eps, out_lin, dense,
                                                      momentum=bn.momentum if training else -1.0, rows_out=act)
            else:
                bstats = ops.enc_bn_relu(new_rows, n_out, n_out_max, stats, bn.weight.detach(), bn.bias.detach(),
                                         bn.running_mean, bn.running_var, training, bn.eps, out_lin, dense,
                                         momentum=bn.momentum if training else -1.0,  # running stats updated in-kernel
                                         rows_out=act)
            if save is not None:
                save.append({"conv": conv, "bn": bn, "stride": stride, "in_rows": rows_f32, "in_grid": grid, "in_dhw": dhw,
                             "in_lin": rows_lin, "n_in": n_rows, "n_in_max": n_max, "out_grid": out_grid,
                             "out_lin": out_lin, "n_out": n_out, "n_out_max": n_out_max, "out_dhw": out_dhw,
                             "x": new_rows, "y": act, "bstats": bstats, "level": len(volumes) - 1 if dense is not None else None})
            if training:
                bn_updates.append(bn.num_batches_tracked)
            rows_f32 = act if save is not None else new_rows
            rows = split if next_split else rows_f32
            grid, rows_lin, n_rows, n_max, dhw = out_grid, out_lin, n_out, n_out_max, out_dhw
            rows_are_split = next_split
        if training:
            torch._foreach_add_(bn_updates, 1)  # nn.BatchNorm1d bookkeeping, one fused launch
        return FeatureVolumes(volumes, sparse)


_MLP_NAMES = {"fc0": "fc_0", "fc1": "fc_1", "fc2": "fc_2", "alpha": "alpha_fc", "feature": "feature_fc",
              "latent": "latent_fc", "view": "view_fc", "rgb": "rgb_fc"}


class Network(nn.Module):
    def __init__(self, num_train_frame, voxel_size=(0.005, 0.005, 0.005), xyz_res=10, view_res=4, precision=None):
        super().__init__()
        # decoder arithmetic: 'f16f6' (fc_0 folded into the volume + fp16 / six-bit cross-term MFMAs), 'f32' (exact fp32
        # MFMA, the reference's precision) or 'auto' = 'f16f6' unless the weights have blocks six bits cannot hold
        self.precision = precision or os.environ.get("NB_PRECISION", DEFAULT_PRECISION)
        if self.precision not in ("auto", "f32", "f16f6"):
            raise ValueError("precision must be 'auto', 'f32' or 'f16f6'")
        self._auto = None  # (weight key, chosen arithmetic, statistic) of precision 'auto'
        self._lb_cache = None  # (latent_index tensor, versions, bias) of latent_bias()
        if int(xyz_res) != 10 or int(view_res) != 4:
            raise NotImplementedError("the HIP decoder is built for xyz_res=10, view_res=4 (view_fc has 346 inputs)")
        self.voxel_size = [float(v) for v in voxel_size]
        self.c = nn.Embedding(N_VERTS, CODE_DIM)
        self.xyzc_net = SparseConvNet()
        self.latent = nn.Embedding(int(num_train_frame), 128)
        self.actvn = nn.ReLU()
        self.fc_0 = nn.Conv1d(352, 256, 1)
        self.fc_1 = nn.Conv1d(256, 256, 1)
        self.fc_2 = nn.Conv1d(256, 256, 1)
        self.alpha_fc = nn.Conv1d(256, 1, 1)
        self.feature_fc = nn.Conv1d(256, 256, 1)
        self.latent_fc = nn.Conv1d(384, 256, 1)
        self.view_fc = nn.Conv1d(346, 128, 1)
        self.rgb_fc = nn.Conv1d(128, 3, 1)
        self._packed = None
        self._packed_key = None
        self._packed_have = set()
        self._t_vals = {}

    # the packed blobs and their keys (storages!) are caches of the parameters: they are neither copied nor pickled
    def __getstate__(self):
        st = dict(self.__dict__)
        st.update(_packed=None, _packed_key=None, _packed_have=set(), _auto=None, _t_vals={}, _lb_cache=None)
        return st

    def __deepcopy__(self, memo):
        import copy

        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            setattr(new, k, copy.deepcopy(v, memo))
        for m in new.modules():
            m.__dict__.pop("_nb_packed16", None)
        return new

    # ------------------------------------------------------------------ packed decoder weights
    def _mlp_param_dict(self):
        d = {}
        for short, name in _MLP_NAMES.items():
            m = getattr(self, name)
            d[short + "_w"] = m.weight
            d[short + "_b"] = m.bias
        return d

    def _point_precision(self):
        """Arithmetic of nb_decode_points (stand-alone points: calculate_density, calculate_density_color, get_pixel_value,
        RendererMesh): the march's (every point is a one-sample ray of the same kernel)."""
        return self.march_precision()

    def march_precision(self):
        """Arithmetic of the fused march.  'auto' = 'f16f6' unless the weights have blocks fp6 cannot hold: more than
        SIX_BIT_MAX_SMALL of a layer's non-zero weights below 1/8 of their (row, 32 K) block maximum (normally distributed
        weights: ~0.2; the wide-dynamic-range stress case of tools/experiments/precision_sweep.py: ~0.75, where six-bit cross
        terms triple the error) — then the exact 'f32' kernel, with a warning (it is ~6x slower).  A function of the weights
        alone — decided once per weight version (one 3-float read-back after packing), the same on every rank, no timing."""
        if self.precision != "auto":
            return self.precision
        packed = self.packed_weights("f16f6")
        if self._auto is None or self._auto[0] is not self._packed_key:
            worst = float(ops.six_bit_small_fraction(packed).max())
            choice = "f16f6" if worst <= SIX_BIT_MAX_SMALL else "f32"
            if choice == "f32":
                import warnings

                warnings.warn("neuralbody_amd: %.0f %% of a decoder layer's weights lie below 1/8 of their 32-wide block maximum; "
                              "precision 'auto' takes the exact fp32 kernel for these weights" % (100 * worst))
            self._auto = (self._packed_key, choice, worst)
        return self._auto[1]

    def packed_weights(self, precision=None):
        """MFMA-ordered decoder blob, rebuilt (on device) whenever a parameter changed; only the sections of the
        arithmetics asked for since the last change are (re)written — a training step repacks every iteration and only
        ever decodes with 'f32'."""
        need = {self.march_precision()} if precision is None else {precision}
        d = self._mlp_param_dict()
        # keyed on the parameters' storages, which the entry keeps alive (so an address cannot be recycled under the
        # key), and on their version counters (optimizer steps and load_state_dict write in place)
        key = tuple((t.untyped_storage(), t.data_ptr(), t._version) for t in d.values())
        old = self._packed_key
        same = old is not None and len(old) == len(key) and all(
            a[0]._cdata == b[0]._cdata and a[1:] == b[1:] for a, b in zip(old, key))
        if self._packed is None or not same:
            self._packed = ops.mlp_pack(d, self._packed, precisions=need)
            self._packed_key = key
            self._packed_have = set(need) | {"f32"}
        elif not need <= self._packed_have:
            ops.mlp_pack(d, self._packed, precisions=need - self._packed_have)
            self._packed_have |= need
        return self._packed

    def latent_bias(self, latent_index):
        """Per-frame bias of the merged feature_fc/latent_fc layer (latent_xyzc.py:108-111)."""
        w = self.latent.weight.detach()
        # The bias is a function of the frame's latent row and four layers' parameters: kept per (latent_index tensor, its
        # version, the parameters' versions) — a view loop re-renders one frame, and the three launches it takes (index_select,
        # copy, nb_mlp_latent_bias: ~40 us) sit between the encoder and the march.  The entry holds the index tensor itself,
        # so its address cannot be recycled under the key; a training step bumps the versions and misses.
        d = None
        if isinstance(latent_index, torch.Tensor):
            d = self._mlp_param_dict()
            vers = (latent_index._version, w._version, w.data_ptr()) + tuple((t.data_ptr(), t._version) for t in d.values())
            old = self._lb_cache
            if old is not None and old[0] is latent_index and old[1] == vers:
                return old[2]
        else:
            latent_index = torch.tensor([int(latent_index)])
        idx = latent_index.reshape(-1)[:1].long().to(w.device)
        row = w.index_select(0, idx)[0].contiguous()
        lb = ops.mlp_latent_bias(self._mlp_param_dict(), row)
        if d is not None:
            self._lb_cache = (latent_index, vers, lb)
        return lb

    # ------------------------------------------------------------------ scene description
    def make_scene(self, feature_volume, sp_input, precision=None):
        """nb_scene of one frame.  R / Th / bounds stay on the device (ops.make_pose packs them into the 15-float
        block the kernels read): no host copy, no sync and — unlike round 1's address-keyed host cache — nothing
        that could hand frame k+1 the pose of frame k when the allocator recycles the batch's addresses.
        precision 'f16f6' marches the fc_0-folded planes of the volumes (ops.fold_build): built once per (volumes, fc_0
        weight version) and kept on the FeatureVolumes object."""
        vols = []
        for v in feature_volume:
            vols.append(v if v.dim() == 4 else ops.volume_as_channels_last(v))
        R, Th, bounds = sp_input["R"], sp_input["Th"], sp_input["bounds"]
        if R.numel() != 9 or bounds.numel() != 6:
            raise NotImplementedError("batch size 1 only (train.batch_size / test batch are 1 in every shipped config)")
        out_sh = [int(s) for s in sp_input["out_sh"]]
        fold = self._fold_planes(feature_volume, vols) if precision == "f16f6" else None
        return ops.make_scene(vols, ops.make_pose(R, Th, bounds, device=vols[0].device), self.voxel_size, out_sh, fold=fold)

    def _fold_planes(self, feature_volume, vols):
        w = self.fc_0.weight.detach()
        key = (w.data_ptr(), w._version)
        fv = feature_volume if isinstance(feature_volume, FeatureVolumes) else None
        if fv is not None and fv.fold is not None and fv.fold[0] == key and fv.fold[1] is w.untyped_storage():
            return fv.fold[2]
        sparse = fv.sparse if fv is not None and fv.sparse is not None else [ops.sparsify(v) for v in vols]
        fold = ops.fold_build(vols, sparse, w)
        if fv is not None:
            fv.fold = (key, w.untyped_storage(), fold)  # holds the storage: its address cannot be recycled under the key
        return fold

    # ------------------------------------------------------------------ reference API
    def encode_sparse_voxels(self, sp_input, save=None):
        coord = sp_input["coord"]
        if int(sp_input.get("batch_size", 1)) != 1:
            raise NotImplementedError("batch size 1 only")
        c3 = sp_input.get("_coord_dhw")  # Renderer.prepare_sp_input: the [n, 3] tensor the [n, 4] one was built from
        if c3 is not None and c3.dim() == 2 and c3.shape == (coord.shape[0], 3) and c3.dtype == torch.int32 and c3.is_contiguous():
            coord = c3
        else:
            if coord.dim() == 2 and coord.shape[1] == 4:  # [N,4] = (batch idx, d, h, w), if_clight_renderer.py:33-38
                coord = coord[:, 1:]
            coord = coord.reshape(-1, 3).to(torch.int32).contiguous()
        codes = self.c.weight.detach()
        vols = self.xyzc_net(codes, coord, sp_input["out_sh"], self.training, save)
        # logical layout [1,C,D,H,W] like spconv's .dense(); storage stays channels-last
        return FeatureVolumes([v.permute(3, 0, 1, 2)[None] for v in vols], vols.sparse)

    SORT_MIN_POINTS = 4096  # below this a spatial sort of the points costs more than it saves

    def _spatial_order(self, p, sp_input):
        """Permutation that makes consecutive points spatial neighbours (index plumbing, no arithmetic of the path): the
        'f16f6' decoder marches 64 consecutive points per workgroup over the union of the voxels they touch, so points in
        ray-major or scan-line order (64 samples of one ray: metres apart) would fall back to one sample at a time.  Key =
        4 cm block of the SMPL-space grid (row-major) x Morton position of the 1 cm voxel inside it; one device sort."""
        R = sp_input["R"].reshape(3, 3).to(p)
        Th = sp_input["Th"].reshape(-1)[:3].to(p)
        bmin = sp_input["bounds"].reshape(-1, 3)[0].to(p)
        q = torch.matmul(p - Th, R)  # latent_xyzc.py:43-46
        vs = torch.tensor([self.voxel_size[2], self.voxel_size[1], self.voxel_size[0]], dtype=p.dtype, device=p.device)
        v = torch.floor((q - bmin) / (2.0 * vs)).clamp_(0, 1023).to(torch.int64)  # level-1 voxel (x, y, z), 10 bits each
        blk = ((v[:, 2] >> 2) << 16) | ((v[:, 1] >> 2) << 8) | (v[:, 0] >> 2)
        x, y, z = v[:, 0] & 3, v[:, 1] & 3, v[:, 2] & 3
        loc = ((z >> 1) << 5) | ((y >> 1) << 4) | ((x >> 1) << 3) | ((z & 1) << 2) | ((y & 1) << 1) | (x & 1)
        return torch.argsort((blk << 6) | loc)

    def _decode(self, wpts, viewdir, feature_volume, sp_input, density_only):
        prec = self._point_precision()
        scene = self.make_scene(feature_volume, sp_input, prec)
        p = wpts.reshape(-1, 3).float().contiguous()
        v = None if density_only else viewdir.reshape(-1, 3).float().contiguous()
        lb = None if density_only else self.latent_bias(sp_input["latent_index"])
        packed = self.packed_weights(prec)
        if prec == "f16f6" and p.shape[0] >= self.SORT_MIN_POINTS:
            order = self._spatial_order(p, sp_input)
            out_s = ops.decode_points(scene, packed, lb, p[order].contiguous(), None if v is None else v[order].contiguous(),
                                      density_only=density_only, precision=prec)
            out = torch.empty_like(out_s)
            out[order] = out_s
            return out
        return ops.decode_points(scene, packed, lb, p, v, density_only=density_only, precision=prec)

    def calculate_density(self, wpts, feature_volume, sp_input):
        if wpts.shape[0] != 1:
            raise NotImplementedError("batch size 1 only")
        return self._decode(wpts, None, feature_volume, sp_input, True).view(1, -1, 1)

    def calculate_density_color(self, wpts, viewdir, feature_volume, sp_input):
        if wpts.shape[0] != 1:
            raise NotImplementedError("batch size 1 only")
        return self._decode(wpts, viewdir, feature_volume, sp_input, False).view(1, -1, 4)

    def forward(self, sp_input, grid_coords, viewdir, light_pts):
        """Working equivalent of the reference's (broken, latent_xyzc.py:128-163) forward: `viewdir`
        [B,N,27] and `light_pts` [B,N,63] are the positional encodings whose first three entries are
        the raw direction / world point (embedder.py:14-17); the volume is sampled at those world
        points, which is where `grid_coords` came from (get_grid_coords)."""
        feature_volume = self.encode_sparse_voxels(sp_input)
        return self.calculate_density_color(light_pts[..., :3], viewdir[..., :3], feature_volume, sp_input)

    # ------------------------------------------------------------------ fused march
    def render_rays(self, ray_o, ray_d, near, far, feature_volume, sp_input, n_samples, t_rand=None,
                    white_bkgd=False, want_raw=False, ray_order=None, cull=None, order_covers_all=False):
        """All rays of the (single) batch element through nb_march.  ray_o/ray_d [n,3], near/far [n]."""
        prec = self.march_precision()
        scene = self.make_scene(feature_volume, sp_input, prec)
        lb = self.latent_bias(sp_input["latent_index"])
        key = (int(n_samples), str(ray_o.device))  # a constant of (S, device), not of the frame
        t_vals = self._t_vals.get(key)
        if t_vals is None:
            t_vals = torch.linspace(0.0, 1.0, steps=int(n_samples)).to(ray_o.device)  # if_clight_renderer.py:13
            self._t_vals[key] = t_vals
        return ops.march(scene, self.packed_weights(prec), lb, ray_o, ray_d, near, far, t_vals, t_rand,
                         white_bkgd=white_bkgd, want_raw=want_raw, precision=prec, ray_order=ray_order,
                         cull=cull, order_covers_all=order_covers_all)
