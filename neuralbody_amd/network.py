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
from collections.abc import Sequence

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
LAZY_DENSE = os.environ.get("NB_LAZY_DENSE", "1") != "0"  # inference: dense volumes materialised on first access only
INDEX_ALL_LEVELS = os.environ.get("NB_INDEX_ALL", "1") != "0"  # the strided levels' index sets in three launches (0: three per level)
SIX_BIT_MAX_SMALL = 0.5  # precision 'auto': largest per-layer share of weights six-bit blocks cannot hold before it takes 'f32'


class FeatureVolumes(Sequence):
    """The four volumes of `Network.encode_sparse_voxels` ([1,C,D,H,W] views of channels-last storage, as the reference's
    `.dense()` returns them, latent_xyzc.py:188-201) together with the index structures they come from: `sparse[l]` = (index grid
    [D,H,W] int32, linear voxel index of every active row, device-side row count [1], row capacity).  A read-only sequence of
    four tensors: `len`, indexing, slicing and iteration work as on the reference's list.  (Deliberately NOT a `list` subclass:
    C-level consumers of lists — `torch.cat`, `PySequence_Fast` — read a list's storage directly and would see the not yet
    materialised volumes as an empty list; handed this object they raise instead.  `list(fv)` / `fv.dense()` give a real list.)

    On the inference path the DENSE tensors are made on first access only (`rows[l]`: the level's active rows, compact fp32
    [capacity, C]; `shapes[l]` = (D, H, W)): the default arithmetic 'f16f6' marches the fc_0-folded planes, which nb_fold_build
    forms from the compact rows, and never reads a dense volume — 137 MB per frame at out_sh (96, 352, 192) that nothing would
    look at.  Indexing, iterating or `dense()` materialises them (zero fill + scatter on the device, no synchronisation):
    `calculate_density(_color)` with precision 'f32', foreign consumers, the tests.  `fold`: (fc_0 weight key, storage,
    ops.fold_build result), rebuilt when the weight changes; a plain list of volumes from elsewhere gets its active set from
    ops.sparsify."""

    def __init__(self, volumes=None, sparse=None, rows=None, shapes=None, zeroed_int=None):
        self._dense = list(volumes) if volumes is not None else None
        self.sparse = sparse
        self.rows = rows
        self.zeroed_int = zeroed_int  # one int32 the encoder's zero fill covered: the first fold build's saturation counter
        if shapes is None and volumes is not None:  # [1,C,D,H,W] views or channels-last [D,H,W,C] storage
            shapes = [tuple(int(x) for x in (v.shape[2:] if v.dim() == 5 else v.shape[:3])) for v in volumes]
        self.shapes = shapes
        self.fold = None
        if self._dense is None and rows is None:
            raise ValueError("FeatureVolumes needs the dense volumes or the levels' compact rows")

    def dense(self):
        """The list of the four [1,C,D,H,W] tensors (materialised once)."""
        if self._dense is None:
            vols = []
            for (grid, rows_lin, n_rows, cap), rows, dhw in zip(self.sparse, self.rows, self.shapes):
                c = int(rows.shape[1])
                nvox = dhw[0] * dhw[1] * dhw[2]
                cap = max(int(cap), 1)
                buf = torch.zeros((nvox + 1, c), dtype=torch.float32, device=rows.device)  # + one row the padding scatters into
                live = torch.arange(cap, device=rows.device) < n_rows
                idx = torch.where(live, rows_lin[:cap].long(), torch.full((), nvox, dtype=torch.int64, device=rows.device))
                buf.index_copy_(0, idx, torch.where(live[:, None], rows[:cap], torch.zeros((), device=rows.device)))
                vols.append(buf[:nvox].view(dhw[0], dhw[1], dhw[2], c).permute(3, 0, 1, 2)[None])
            self._dense = vols
        return self._dense

    def is_dense(self):
        return self._dense is not None

    def __len__(self):
        return len(self.shapes) if self._dense is None else len(self._dense)

    def __getitem__(self, i):
        return self.dense()[i]

    def __iter__(self):
        return iter(self.dense())


class BatchedFeatureVolumes(Sequence):
    """`encode_sparse_voxels` of a batch of B > 1 frames: a read-only sequence of four [B,C,D,H,W] tensors like the reference's list
    (concatenated from the frames' volumes on first access), with `frames[b]` = frame b's own FeatureVolumes, which is what this
    package's render / decode paths take (no [B, ...] tensor is made for them)."""

    def __init__(self, frames):
        self.frames = list(frames)
        self._dense = None

    def dense(self):
        if self._dense is None:
            self._dense = [torch.cat([f[l] for f in self.frames], 0) for l in range(4)]
        return self._dense

    def __len__(self):
        return 4

    def __getitem__(self, i):
        return self.dense()[i]

    def __iter__(self):
        return iter(self.dense())


def frame_sp_input(sp_input, b):
    """The sp_input of batch element b alone (batch_size 1): its rows of `coord` (equal counts per element, as
    if_clight_renderer.py:33-38 builds them; batch-index column 0), its bounds / R / Th / latent_index, the batch's common out_sh."""
    B = int(sp_input.get("batch_size", 1))
    coord = sp_input["coord"]
    n = coord.shape[0] // B
    c = coord[b * n:(b + 1) * n]
    if c.dim() == 2 and c.shape[1] == 4:
        c = torch.cat([torch.zeros_like(c[:, :1]), c[:, 1:]], dim=1)
    out = {"coord": c, "out_sh": list(sp_input["out_sh"]), "batch_size": 1}
    for k in ("bounds", "R", "Th", "latent_index"):
        if k in sp_input:
            out[k] = sp_input[k][b:b + 1]
    return out


def frame_volumes(feature_volume, b):
    """Frame b's volumes out of what encode_sparse_voxels returned for a batch (or out of a plain list of [B,C,D,H,W] tensors)."""
    if isinstance(feature_volume, BatchedFeatureVolumes):
        return feature_volume.frames[b]
    return [v[b:b + 1] for v in feature_volume]


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

    @staticmethod
    def _wkey(conv):
        w = conv.weight.detach()
        return w, (w.untyped_storage(), w.data_ptr(), w._version)

    @staticmethod
    def _fresh(old, key):
        return old is not None and old[0][0]._cdata == key[0]._cdata and old[0][1:] == key[1:]

    def _packed16(self, conv, backward_input=False):
        """fp16 head / remainder B fragments of one convolution's weight (backward_input: the bf16 pairs of its backward-input
        convolution), rebuilt when the parameter changes (the entry holds the storage it was packed from, so its address cannot
        be recycled under the key)."""
        w, key = self._wkey(conv)
        attr = "_nb_packed16_bwd" if backward_input else "_nb_packed16"
        old = getattr(conv, attr, None)
        if not self._fresh(old, key):
            old = (key, ops.enc_conv_pack16(w, backward_input=backward_input))
            setattr(conv, attr, old)
        return old[1]

    def repack_stale(self, with_backward):
        """Every stale packed form of the >= 32-channel convolutions in ONE launch (after an optimiser step all of them are stale:
        14 forward forms + 14 backward-input forms = 28 launches otherwise)."""
        jobs, slots = [], []
        for name, cin, cout, n, stride in ENCODER_BLOCKS:
            block = getattr(self, name)
            for j in range(n):
                conv = block[3 * j]
                if int(conv.weight.shape[3]) < 32:
                    continue
                w, key = self._wkey(conv)
                forms = [("_nb_packed16", False)] + ([("_nb_packed16_bwd", True)] if with_backward else [])
                for attr, bwd in forms:
                    if not self._fresh(getattr(conv, attr, None), key):
                        jobs.append((w, bwd))
                        slots.append((conv, attr, key))
        if jobs:
            for (conv, attr, key), packed in zip(slots, ops.enc_conv_pack16_batch(jobs)):
                setattr(conv, attr, (key, packed))

    def forward(self, codes, coord, out_sh, training, save=None, dense=True):
        """codes [6890,16] fp32, coord [6890,3] int32 (d,h,w) -> 4 channels-last volumes [D,H,W,C].
        save: optional list that receives one record per conv+BN+ReLU layer (index structures, raw and activated
        rows, batch statistics) — everything neuralbody_amd.training.encoder_backward needs.
        dense=False (inference): the `.dense()` volumes of latent_xyzc.py:188-201 are neither zero-filled nor scattered; the
        result carries the levels' active rows instead (FeatureVolumes.rows) and materialises the volumes on first access."""
        lazy = not dense and save is None
        dev = codes.device
        dhw = [int(s) for s in out_sh]
        n_max = coord.shape[0]
        layers = [(name, cin, cout, n, stride, j) for name, cin, cout, n, stride in ENCODER_BLOCKS for j in range(n)]
        # ONE zero fill for everything the pass needs cleared: the index buffers of the levels (rows_vert | rows_lin | n_rows, then
        # out_lin | n_out per strided layer), the layers' fp64 statistics and (when made eagerly) the dense volumes
        int_sizes, dense_shapes, cap, d = [1, 2 * max(n_max, 1) + 1], [], n_max, dhw
        for name, cin, cout, n, stride, j in layers:
            if stride == 2:
                int_sizes.append(ops.down_capacity(cap, d) + 1)
                cap, d = int_sizes[-1] - 1, ops.down_dhw(d)
            if name in DENSE_AFTER and j == n - 1:
                dense_shapes.append(d + [cout])
        n_int = (sum(int_sizes) + 63) // 64 * 64  # (the fp32 / fp64 part behind it keeps a fresh allocation's 256-byte alignment)
        # ... and ONE fill with -1 for the index grids of the five levels
        grid_shapes, d = [list(dhw)], dhw
        for name, cin, cout, n, stride, j in layers:
            if stride == 2:
                d = ops.down_dhw(d)
                grid_shapes.append(list(d))
        grid_sizes = [(math.prod(sh) + 63) // 64 * 64 for sh in grid_shapes]
        grid_bufs = [b[:math.prod(sh)].view(sh) for b, sh in zip(torch.full((sum(grid_sizes),), -1, dtype=torch.int32, device=dev).split(grid_sizes),
                                                                 grid_shapes)]
        n_stats = 2 * len(layers) * 256  # fp64 [layers, 256] in front (8-byte aligned), the volumes behind it (64-float aligned)
        dense_sizes = [0 if lazy else (math.prod(sh) + 63) // 64 * 64 for sh in dense_shapes]
        # ONE zero fill: the index buffers (int32), the layers' statistics (fp64) and, when they are made eagerly, the volumes — an
        # all-zero word is 0 in every one of these types
        zero_buf = torch.zeros(n_int + n_stats + sum(dense_sizes), dtype=torch.int32, device=dev)
        int_bufs = list(zero_buf[:sum(int_sizes)].split(int_sizes))
        zeroed_int = int_bufs.pop(0)
        f32_buf = zero_buf[n_int:].view(torch.float32)
        stats_all = f32_buf[:n_stats].view(torch.float64).view(len(layers), 256)
        dense_bufs = [None if lazy else b[:math.prod(sh)].view(sh) for b, sh in zip(f32_buf[n_stats:].split(dense_sizes), dense_shapes)]
        level_rows, level_shapes = [], []
        grid, rows_vert, rows_lin, n_rows = ops.enc_voxelize(coord, dhw, buf=int_bufs.pop(0), grid=grid_bufs.pop(0))
        # the index sets of all strided levels at once (three launches; a level at a time costs three each): they follow from the
        # voxelised level alone
        n_strided = sum(1 for layer in layers if layer[4] == 2)
        down_sets = ops.enc_downsample_index_all(rows_lin, n_rows, n_max, dhw, int_bufs[:n_strided], grid_bufs[:n_strided]) if INDEX_ALL_LEVELS and 1 <= n_strided <= 4 else None
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
        if fast:
            self.repack_stale(with_backward=save is not None)
        rows_are_split = False
        rows_f32 = rows  # the fp32 form of the current layer's input rows (what the backward record keeps)
        for li, (name, cin, cout, n, stride, j) in enumerate(layers):
            block = getattr(self, name)
            conv, bn = block[3 * j], block[3 * j + 1]
            if stride == 2 and down_sets is not None:
                out_grid, out_lin, n_out, n_out_max, out_dhw = down_sets.pop(0)
            elif stride == 2:
                out_grid, out_lin, n_out, n_out_max, out_dhw = ops.enc_downsample_index(rows_lin, n_rows, n_max, dhw, buf=int_bufs.pop(0),
                                                                                           grid=grid_bufs.pop(0))
            else:
                out_grid, out_lin, n_out, n_out_max, out_dhw = grid, rows_lin, n_rows, n_max, dhw
            if rows_are_split:
                new_rows, stats = ops.enc_conv16(rows, grid, dhw, out_lin, n_out, n_out_max, out_dhw, stride,
                                                 self._packed16(conv), cin, cout, stats=stats_all[li, :2 * cout])
            else:
                new_rows, stats = ops.enc_conv(rows, grid, dhw, out_lin, n_out, n_out_max, out_dhw, stride,
                                               conv.weight.detach(), stats=stats_all[li, :2 * cout])
            dense = None
            is_level = name in DENSE_AFTER and j == n - 1
            if is_level:
                dense = dense_bufs.pop(0)
                assert lazy or list(dense.shape) == out_dhw + [cout]
                volumes.append(dense)
                sparse.append((out_grid, out_lin, n_out, n_out_max))
                level_shapes.append(tuple(out_dhw))
            next_split = fast and li + 1 < len(layers) and cout >= 32  # the consumer of these rows is an enc_conv16
            # the activated rows in fp32 beside the raw conv output: when saving (the backward needs both), and for a level
            # whose dense volume is not written (its consumer convolution takes the split planes, nb_fold_build these rows)
            act = torch.empty_like(new_rows) if (save is not None or (lazy and is_level and next_split)) else None
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
            if lazy and is_level:
                level_rows.append(act if act is not None else new_rows)  # (bn_relu without rows_out activates in place)
            rows_f32 = act if save is not None else new_rows
            rows = split if next_split else rows_f32
            grid, rows_lin, n_rows, n_max, dhw = out_grid, out_lin, n_out, n_out_max, out_dhw
            rows_are_split = next_split
        if training:
            torch._foreach_add_(bn_updates, 1)  # nn.BatchNorm1d bookkeeping, one fused launch
        if lazy:
            return FeatureVolumes(None, sparse, rows=level_rows, shapes=level_shapes, zeroed_int=zeroed_int)
        return FeatureVolumes(volumes, sparse, zeroed_int=zeroed_int)


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
        # 'f16f6': the march lists the rays whose LAST density it cannot sign (the reference's 1e10 interval makes that sample's alpha
        # a step function, nerf_net_utils.py:28) and recomputes those at fp32 level (include/nb_hip.h, nb_march `ill_scratch`)
        self.last_sample_fixup = os.environ.get("NB_LAST_SAMPLE_FIXUP", "1") != "0"
        self._auto = None  # (weight key, chosen arithmetic, statistic) of precision 'auto'
        self._lb_cache = None  # (latent_index tensor, versions, bias) of latent_bias()
        self._foreign_fold = None  # (volume tensors + versions, fc_0 key, storage, planes) of volumes that came as a plain list
        self._sat_checked = None  # fc_0 weight key whose first fold build had its saturation count read (precision 'auto')
        self._planes_overflow = None  # ... and, if that count was not zero, the key again: 'auto' = 'f32' for these weights
        self.last_ill = None
        self._pose_cache = None  # (R, Th, bounds tensors, versions, the 15-float pose block) of the last make_scene
        self._sat_pending = None  # (fc_0 key, counter tensor) of a later frame's planes, read at the next host synchronisation
        if int(xyz_res) != 10 or int(view_res) != 4:
            raise NotImplementedError(
                "xyz_res=10, view_res=4 only: the reference's own Network hard-codes view_fc = Conv1d(346, 128, 1) "
                "(latent_xyzc.py:27: 256 + 27 + 63 = 346 inputs), so any other embedder resolution (embedder.py:53-54) fails there "
                "too, at the first view_fc call (:120); the HIP decoder is built for the same 346 inputs")
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
        st.update(_packed=None, _packed_key=None, _packed_have=set(), _auto=None, _t_vals={}, _lb_cache=None, _foreign_fold=None,
                  _sat_checked=None, _planes_overflow=None, _sat_pending=None, last_ill=None, _pose_cache=None)
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
            m.__dict__.pop("_nb_packed16_bwd", None)
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
        if self._planes_overflow is not None:
            w = self.fc_0.weight
            if self._planes_overflow == (w.data_ptr(), w._version):
                return "f32"  # fc_0 . V does not fit the fp16 planes (nb_fold_build's count, _auto_checks_planes)
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
        lazy = isinstance(feature_volume, FeatureVolumes) and not feature_volume.is_dense()
        if lazy and precision == "f16f6":
            vols = list(feature_volume.shapes)  # the folded planes are all this arithmetic reads: no dense volume is made
        else:
            vols = [v if v.dim() == 4 else ops.volume_as_channels_last(v) for v in feature_volume]
        R, Th, bounds = sp_input["R"], sp_input["Th"], sp_input["bounds"]
        if R.numel() != 9 or bounds.numel() != 6:
            raise ValueError("make_scene describes ONE frame: R / Th / bounds of a batch go through frame_sp_input (Renderer.render and "
                             "calculate_density(_color) loop the batch)")
        out_sh = [int(s) for s in sp_input["out_sh"]]
        fold = self._fold_planes(feature_volume, vols) if precision == "f16f6" else None
        dev = feature_volume.rows[0].device if lazy else vols[0].device
        return ops.make_scene(vols, self._pose_block(R, Th, bounds, dev), self.voxel_size, out_sh, fold=fold)

    def _pose_block(self, R, Th, bounds, dev):
        """ops.make_pose, kept per (R, Th, bounds) tensor objects and versions: a loop over the views of one frame (and the two
        make_scene calls of a prefetched render) builds the 15-float block once instead of one concatenation launch each time.  The
        entry holds the tensors themselves, so an address cannot be recycled under the key (tests/test_gpu_frames.py)."""
        c = self._pose_cache
        if (c is not None and all(isinstance(t, torch.Tensor) for t in (R, Th, bounds)) and c[0] is R and c[1] is Th and c[2] is bounds
                and c[3] == (R._version, Th._version, bounds._version, str(dev))):
            return c[4]
        pose = ops.make_pose(R, Th, bounds, device=dev)
        if all(isinstance(t, torch.Tensor) for t in (R, Th, bounds)):
            self._pose_cache = (R, Th, bounds, (R._version, Th._version, bounds._version, str(dev)), pose)
        return pose

    def _fold_planes(self, feature_volume, vols):
        """(NbFold, keepalive) of the volumes: from the encoder's compact rows when the FeatureVolumes object carries them (no
        dense volume is touched), from its dense volumes + index structures otherwise; a plain list of dense volumes gets its
        active set from ops.sparsify, sized by ONE read-back of the four counts and kept per (volume tensors, versions, fc_0)."""
        w = self.fc_0.weight.detach()
        key = (w.data_ptr(), w._version)
        fv = feature_volume if isinstance(feature_volume, FeatureVolumes) else None
        if fv is not None and fv.fold is not None and fv.fold[0] == key and fv.fold[1] is w.untyped_storage():
            return fv.fold[2]
        if fv is not None and fv.sparse is not None:
            n_sat, fv.zeroed_int = fv.zeroed_int, None  # (a rebuild for a new fc_0 gets a fresh counter)
            if fv.rows is not None:
                fold = ops.fold_build(fv.shapes, fv.sparse, w, rows=fv.rows, n_sat=n_sat)
            else:
                fold = ops.fold_build(vols, fv.sparse, w, n_sat=n_sat)
            fv.fold = (key, w.untyped_storage(), fold)  # holds the storage: its address cannot be recycled under the key
            self._auto_checks_planes(key, fold)
            return fold
        # volumes from elsewhere (cloned, loaded, another encoder's): nothing to hang a cache on but the tensors themselves
        # (`vols` are fresh channels-last VIEWS of the caller's tensors at every call: the key is their storage, address and version;
        # the entry keeps the views, hence the storages, alive, so an address cannot be recycled under it)
        fkey = tuple((v, v.untyped_storage()._cdata, v.data_ptr(), v._version, tuple(v.shape)) for v in vols)
        old = self._foreign_fold
        if old is not None and old[1] == key and old[2] is w.untyped_storage() and len(old[0]) == len(fkey) and \
                all(a[1:] == b[1:] for a, b in zip(old[0], fkey)):
            return old[3]
        sparse = [ops.sparsify(v) for v in vols]
        counts = torch.cat([sp[2] for sp in sparse]).tolist()  # one read-back: a capacity of every voxel would be ~1 GB of planes
        sparse = [(g, lin, n, max(int(c), 1)) for (g, lin, n, _), c in zip(sparse, counts)]
        fold = ops.fold_build(vols, sparse, w)
        self._foreign_fold = (fkey, key, w.untyped_storage(), fold)
        self._auto_checks_planes(key, fold)
        return fold

    def _auto_checks_planes(self, key, fold):
        """precision 'auto', once per fc_0 version: read the planes' saturation count (one 4-byte read-back); a non-zero count
        means fc_0 . V left the fp16 range somewhere, and this Network takes the exact kernel from here on (with a warning)."""
        if self.precision != "auto":
            return
        if self._sat_checked == key:
            # a later frame of the same weights: saturation depends on the frame's volumes too, but reading its counter here would
            # drain the launch queue once per frame.  The counter is parked instead and read where the host waits for the device
            # anyway (`check_pending_saturation`, called by Renderer at its per-frame out_sh read-back): a frame whose products
            # leave the fp16 range is marched clamped ONCE, then 'auto' warns and takes 'f32' for these weights.
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(fold[1][1].device))  # the stream that is building these planes (maybe the prefetch one)
            self._sat_pending = (key, fold[1][1], ev)
            return
        self._sat_checked = key
        self._note_saturation(key, int(fold[1][1]))

    def check_pending_saturation(self):
        """Read the saturation counter of the last fold planes built since the previous check (one 4-byte read-back; call it
        where the host synchronises with the device anyway).  Returns the count (0: nothing pending or nothing saturated)."""
        pend = getattr(self, "_sat_pending", None)
        if pend is None or self.precision != "auto" or not pend[2].query():  # (planes still being built: look again next time)
            return 0
        self._sat_pending = None
        n = int(pend[1])
        self._note_saturation(pend[0], n)
        return n

    def _note_saturation(self, key, n):
        if n:
            import warnings

            warnings.warn("neuralbody_amd: %d products of fc_0 with the latent volumes exceed the fp16 range of the folded planes; "
                          "precision 'auto' takes the exact fp32 kernel for these weights" % n)
            self._planes_overflow = key

    def fold_saturated(self, feature_volume):
        """How many fc_0 . V products of the volumes' folded planes left the fp16 range or are not finite (device -> host read;
        0 for any sane weights: the planes then carry fc_0 . V to fp32 accuracy).  Non-zero: use precision 'f32'."""
        fv = feature_volume if isinstance(feature_volume, FeatureVolumes) else None
        fold = fv.fold[2] if fv is not None and fv.fold is not None else (self._foreign_fold[3] if self._foreign_fold else None)
        return 0 if fold is None else int(fold[1][1])

    # ------------------------------------------------------------------ reference API
    def encode_sparse_voxels(self, sp_input, save=None):
        B = int(sp_input.get("batch_size", 1))
        c3 = sp_input.get("_coord_dhw") if B == 1 else None  # Renderer.prepare_sp_input: the [n, 3] tensor the [n, 4] one is built from
        coord = c3 if c3 is not None else sp_input["coord"]  # (indexing 'coord' is what makes a lazily built sp_input concatenate it)
        if B != 1:
            # B > 1 frames = B independent passes (BatchNorm statistics per frame).  The reference itself cannot run this case:
            # it pairs ONE set of 6890 codes with the B * 6890 coordinates (latent_xyzc.py:35-36), which spconv indexes out of
            # bounds — so the only defined semantics are its own B = 1 results, frame by frame
            if save is not None:
                raise NotImplementedError("the differentiable path encodes one frame per pass (Renderer.render loops the batch)")
            if coord.dim() != 2 or coord.shape[0] % B:
                raise ValueError("encode_sparse_voxels: coord %s does not hold equal row counts for %d frames" % (tuple(coord.shape), B))
            return BatchedFeatureVolumes([self.encode_sparse_voxels(frame_sp_input(sp_input, b)) for b in range(B)])
        if c3 is not None and c3.dim() == 2 and c3.shape[1] == 3 and c3.dtype == torch.int32 and c3.is_contiguous():
            coord = c3
        else:
            coord = sp_input["coord"]
            if coord.dim() == 2 and coord.shape[1] == 4:  # [N,4] = (batch idx, d, h, w), if_clight_renderer.py:33-38
                coord = coord[:, 1:]
            coord = coord.reshape(-1, 3).to(torch.int32).contiguous()
        codes = self.c.weight.detach()
        if save is None and LAZY_DENSE:
            # inference: the dense volumes are made when (and if) somebody looks at them (FeatureVolumes)
            return self.xyzc_net(codes, coord, sp_input["out_sh"], self.training, None, dense=False)
        vols = self.xyzc_net(codes, coord, sp_input["out_sh"], self.training, save)
        # logical layout [1,C,D,H,W] like spconv's .dense(); storage stays channels-last
        return FeatureVolumes([v.permute(3, 0, 1, 2)[None] for v in vols], vols.sparse, zeroed_int=vols.zeroed_int)

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

    def _decode(self, wpts, viewdir, feature_volume, sp_input, density_only, precision=None):
        prec = precision or self._point_precision()
        scene = self.make_scene(feature_volume, sp_input, prec)
        if precision is None and prec != self._point_precision():  # (see render_rays)
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

    def calculate_density(self, wpts, feature_volume, sp_input, precision=None):
        """latent_xyzc.py:74-89.  precision (an extension): 'f32' / 'f16f6' for this call instead of the Network's arithmetic."""
        B = wpts.shape[0]
        if B != 1:  # frame by frame: every element has its own volumes, pose and bounds
            return torch.cat([self.calculate_density(wpts[b:b + 1], frame_volumes(feature_volume, b), frame_sp_input(dict(sp_input, batch_size=B), b),
                                                     precision=precision) for b in range(B)], 0)
        return self._decode(wpts, None, feature_volume, sp_input, True, precision).view(1, -1, 1)

    LAST_DENSITY_FIX_RAYS = 4096  # rays per frame whose last density the unfused path re-decodes at fp32 (the nearest to zero)

    def fix_last_densities(self, raw, wpts, feature_volume, sp_input):
        """The unfused path's counterpart of nb_march's last-sample fix-up (include/nb_hip.h, `ill_scratch`): raw [B, n, S, 4] as decoded
        by calculate_density_color with the 'f16f6' arithmetic, wpts [B, n, S, 3].  The reference gives a ray's LAST sample the interval
        1e10 (nerf_net_utils.py:28), so its alpha is a step function of the sign of its density: the LAST_DENSITY_FIX_RAYS rays of a frame
        whose last density is nearest to zero are decoded once more with the exact kernel, and those within NB_ILL_SIGMA of zero take the
        exact value.  No host read-back (a fixed number of rays: top-k on the device).  Identity for the 'f32' arithmetic."""
        from ._lib import ILL_SIGMA

        if self._point_precision() != "f16f6" or not self.last_sample_fixup or raw.shape[1] == 0:
            return raw
        raw = raw.clone()
        for b in range(raw.shape[0]):
            sig = raw[b, :, -1, 3]
            k = min(int(sig.shape[0]), self.LAST_DENSITY_FIX_RAYS)
            idx = torch.topk(sig.abs(), k, largest=False).indices
            fv = frame_volumes(feature_volume, b) if raw.shape[0] > 1 else feature_volume
            sp = frame_sp_input(dict(sp_input, batch_size=raw.shape[0]), b) if raw.shape[0] > 1 else sp_input
            exact = self.calculate_density(wpts[b, idx, -1][None].contiguous(), fv, sp, precision="f32")[0, :, 0]
            raw[b, idx, -1, 3] = torch.where(sig[idx].abs() < ILL_SIGMA, exact, sig[idx])
        return raw

    def calculate_density_color(self, wpts, viewdir, feature_volume, sp_input):
        B = wpts.shape[0]
        if B != 1:
            return torch.cat([self.calculate_density_color(wpts[b:b + 1], viewdir[b:b + 1], frame_volumes(feature_volume, b),
                                                           frame_sp_input(dict(sp_input, batch_size=B), b)) for b in range(B)], 0)
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
        if prec != self.march_precision():  # 'auto' just learnt that these weights overflow the folded planes
            prec = self.march_precision()
            scene = self.make_scene(feature_volume, sp_input, prec)
        lb = self.latent_bias(sp_input["latent_index"])
        key = (int(n_samples), str(ray_o.device))  # a constant of (S, device), not of the frame
        t_vals = self._t_vals.get(key)
        if t_vals is None:
            t_vals = torch.linspace(0.0, 1.0, steps=int(n_samples)).to(ray_o.device)  # if_clight_renderer.py:13
            self._t_vals[key] = t_vals
        ret = ops.march(scene, self.packed_weights(prec), lb, ray_o, ray_d, near, far, t_vals, t_rand,
                        white_bkgd=white_bkgd, want_raw=want_raw, precision=prec, ray_order=ray_order,
                        cull=cull, order_covers_all=order_covers_all, fixup=self.last_sample_fixup)
        # the scratch of the march's last-sample fix-up ('f16f6'; include/nb_hip.h `ill_scratch`): int32 [0] = rays listed,
        # [1] = rays whose last alpha changed side — diagnostics for bench.py / the tests, never read back by the product path
        self.last_ill = ret.pop("ill_scratch", None)
        return ret
