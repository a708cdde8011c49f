"""Two independent restatements of the spconv 1.x layers of the reference's encoder (lib/networks/latent_xyzc.py:208-274) held
against each other: `oracle/spconv_standin.py` (dense masked conv3d: what every fixture runs on) and `oracle/spconv_rulebook.py`
(hash table -> rulebook -> gather / GEMM / scatter, how spconv computes).  spconv v1.2.1 itself is absent (INSTALL.md:15-22):
this does not pin parity with it, it only shows that the published semantics were read the same way twice — duplicates, voxels on
the border, odd spatial sizes under stride 2, the [kD, kH, kW, Cin, Cout] weight layout, BatchNorm over active rows."""
import numpy as np
import pytest
import torch

from oracle import spconv_rulebook as rb
from oracle import spconv_standin as sp


def _case(seed, shape, n, cin, dup=0, border=True):
    rs = np.random.RandomState(seed)
    D, H, W = shape
    idx = np.stack([np.zeros(n, np.int64), rs.randint(0, D, n), rs.randint(0, H, n), rs.randint(0, W, n)], 1)
    if border:  # corners and faces of the grid: neighbours outside must count as inactive
        idx[:4] = [[0, 0, 0, 0], [0, D - 1, H - 1, W - 1], [0, 0, H - 1, 0], [0, D - 1, 0, W - 1]]
    if dup:
        idx[-dup:] = idx[:dup]  # the same voxels again, with other features: the later rows win
    feats = rs.randn(n, cin)
    return idx, feats


def _standin_dense(layer_cls, idx, feats, shape, weight, **kw):
    x = sp.SparseConvTensor(torch.from_numpy(feats), torch.from_numpy(idx), shape, 1)
    layer = layer_cls(weight.shape[3], weight.shape[4], 3, bias=False, **kw).double()
    with torch.no_grad():
        layer.weight.copy_(torch.from_numpy(weight))
        y = layer(x)
    return y.dense().numpy(), y.mask.numpy()[:, 0], y.spatial_shape


@pytest.mark.parametrize("shape,n,cin,cout,dup", [((6, 7, 5), 40, 4, 6, 0), ((9, 4, 8), 70, 3, 5, 9), ((3, 3, 3), 27, 2, 2, 0)])
def test_submanifold_convolution_two_ways(shape, n, cin, cout, dup):
    idx, feats = _case(1, shape, n, cin, dup)
    weight = np.random.RandomState(2).randn(3, 3, 3, cin, cout)
    want, mask, _ = _standin_dense(sp.SubMConv3d, idx, feats, shape, weight)
    keys, pairs = rb.subm_rulebook(idx, shape)
    got = rb.dense(keys, rb.apply_rulebook(feats, weight, len(keys), pairs), 1, shape)
    act = np.zeros((1,) + tuple(shape), bool)
    act[keys[:, 0], keys[:, 1], keys[:, 2], keys[:, 3]] = True
    assert np.array_equal(act, mask)
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("shape,n,cin,cout,dup", [((6, 8, 4), 30, 4, 6, 0), ((7, 9, 5), 50, 3, 5, 6), ((5, 5, 5), 8, 2, 3, 0), ((11, 6, 13), 90, 2, 4, 0)])
def test_strided_convolution_two_ways(shape, n, cin, cout, dup):
    """stride 2, padding 1 on even AND odd sizes: output size (in + 2 - 3) // 2 + 1, active where the 3^3 receptive field holds an
    active input (the stand-in: max_pool3d of the mask)."""
    idx, feats = _case(3, shape, n, cin, dup)
    weight = np.random.RandomState(4).randn(3, 3, 3, cin, cout)
    want, mask, out_shape = _standin_dense(sp.SparseConv3d, idx, feats, shape, weight, stride=2, padding=1)
    keys, rb_shape, pairs = rb.sparse_rulebook(idx, shape)
    assert list(rb_shape) == list(out_shape) == [(s - 1) // 2 + 1 for s in shape]
    got = rb.dense(keys, rb.apply_rulebook(feats, weight, len(keys), pairs), 1, rb_shape)
    act = np.zeros((1,) + tuple(rb_shape), bool)
    act[keys[:, 0], keys[:, 1], keys[:, 2], keys[:, 3]] = True
    assert np.array_equal(act, mask)
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())


def test_weight_layout_is_kd_kh_kw_cin_cout():
    """One non-zero weight W[kz, ky, kx, ci, co] moves channel ci of the voxel at offset (kz - 1, ky - 1, kx - 1) into channel co: the
    axis order of the kernel is (z, y, x) like the indices, and the last two axes are (in, out)."""
    shape = (5, 5, 5)
    idx = np.array([[0, 2, 2, 2], [0, 3, 2, 1]], np.int64)  # the second voxel = the first + (dz, dy, dx) = (+1, 0, -1)
    feats = np.array([[1.0, 10.0], [100.0, 1000.0]])
    weight = np.zeros((3, 3, 3, 2, 3))
    weight[2, 1, 0, 1, 2] = 1.0  # offset (+1, 0, -1), input channel 1 -> output channel 2
    for dense_out in (_standin_dense(sp.SubMConv3d, idx, feats, shape, weight)[0],
                      rb.dense(*(lambda k, p: (k, rb.apply_rulebook(feats, weight, len(k), p)))(*rb.subm_rulebook(idx, shape)), 1, shape)):
        assert dense_out[0, 2, 2, 2, 2] == 1000.0  # out[first voxel] takes in[first + offset] = the second voxel's channel 1
        assert dense_out[0, :, 3, 2, 1].sum() == 0.0 and np.count_nonzero(dense_out) == 1


def test_block_of_the_encoder_two_ways():
    """down-conv (stride 2) -> BN over active rows -> ReLU -> two submanifold convs with BN / ReLU -> .dense(): the shape of one
    encoder stage (latent_xyzc.py:184-205), both formulations, float64."""
    shape = (9, 7, 10)
    idx, feats = _case(5, shape, 60, 4, dup=5)
    rs = np.random.RandomState(6)
    ws = [rs.randn(3, 3, 3, 4, 6), rs.randn(3, 3, 3, 6, 6), rs.randn(3, 3, 3, 6, 6)]
    gb = [(rs.rand(6) + 0.5, rs.randn(6) * 0.1) for _ in range(3)]
    # stand-in
    x = sp.SparseConvTensor(torch.from_numpy(feats), torch.from_numpy(idx), shape, 1)
    mods = []
    for i, w in enumerate(ws):
        conv = (sp.SparseConv3d(w.shape[3], w.shape[4], 3, 2, padding=1, bias=False) if i == 0 else sp.SubMConv3d(w.shape[3], w.shape[4], 3, bias=False)).double()
        bn = torch.nn.BatchNorm1d(6, eps=1e-3, momentum=0.01).double()
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(w))
            bn.weight.copy_(torch.from_numpy(gb[i][0]))
            bn.bias.copy_(torch.from_numpy(gb[i][1]))
        mods += [conv, bn, torch.nn.ReLU()]
    seq = sp.SparseSequential(*mods).train()
    with torch.no_grad():
        want = seq(x).dense().numpy()
    # rulebook
    keys, out_shape, pairs = rb.sparse_rulebook(idx, shape)
    rows = rb.batchnorm_relu_rows(rb.apply_rulebook(feats, ws[0], len(keys), pairs), *gb[0])
    for i in (1, 2):
        k2, p2 = rb.subm_rulebook(keys, out_shape)
        assert np.array_equal(k2, keys)
        rows = rb.batchnorm_relu_rows(rb.apply_rulebook(rows, ws[i], len(keys), p2), *gb[i])
    got = rb.dense(keys, rows, 1, out_shape)
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-10 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("shape", [(61, 90, 47), (16, 16, 16), (17, 9, 30), (2, 1, 2), (5, 4, 3)])
def test_active_sets_of_stacked_strided_levels_follow_from_the_base_level_alone(shape):
    """The rule nb_enc_downsample_index_all marks with: a cell c of the l-th k=3 / s=2 / p=1 level below a voxel set is active iff an
    active base voxel p lies within [2^l c - (2^l - 1), 2^l c + (2^l - 1)] in every coordinate, i.e. c in {p >> l, (p + 2^l - 1) >> l}
    per axis (clipped to the level's grid) — against the level-by-level rule of the reference's stack (max_pool3d(3, 2, 1) of the
    occupancy, which is what SparseConv3d's output index set is: oracle/neuralbody_oracle.py), odd and even sizes, four levels."""
    rs = np.random.RandomState(sum(shape))
    n = max(3, int(0.02 * np.prod(shape)))
    pts = np.stack([rs.randint(0, s, size=n) for s in shape], 1)
    pts[0] = [s - 1 for s in shape]
    pts[1] = 0
    occ = np.zeros(shape, dtype=bool)
    occ[tuple(pts.T)] = True
    chained = torch.from_numpy(occ)[None, None].float()
    dims = list(shape)
    for level in range(1, 5):
        chained = (torch.nn.functional.max_pool3d(chained, 3, 2, 1) > 0).float()
        dims = [(d - 1) // 2 + 1 for d in dims]
        assert list(chained.shape[2:]) == dims
        direct = np.zeros(dims, dtype=bool)
        up = (1 << level) - 1
        for p in np.argwhere(occ):
            cells = [sorted({int(v) >> level, (int(v) + up) >> level}) for v in p]
            for cz in cells[0]:
                for cy in cells[1]:
                    for cx in cells[2]:
                        if cz < dims[0] and cy < dims[1] and cx < dims[2]:
                            direct[cz, cy, cx] = True
        assert np.array_equal(direct, chained[0, 0].numpy() > 0), level


@pytest.mark.parametrize("shape", [(9, 7, 10), (8, 8, 8), (5, 4, 3)])
def test_backward_input_of_a_strided_layer_is_a_convolution_with_a_transposed_gather(shape):
    """What nb_enc_conv16(stride = -2) with a mode-1 weight pack computes — input voxel p takes, under MIRRORED offset k', the output voxel
    (p - 1 + k') / 2 where that division is exact in all three coordinates, times W[26 - o']^T — against the transpose of the layer's
    own rulebook (d in[src] += d out[q] . W[o]^T over the pairs of every offset), float64."""
    idx, _ = _case(11, shape, 50, 3)
    rs = np.random.RandomState(12)
    w = rs.randn(3, 3, 3, 3, 5)
    table = rb._hash_rows(idx)
    keys, out_shape, pairs = rb.sparse_rulebook(idx, shape)
    dout = rs.randn(len(keys), 5)
    want = np.zeros((len(idx), 3))
    wk = w.reshape(27, 3, 5)
    for o, pr in enumerate(pairs):
        if len(pr):
            np.add.at(want, pr[:, 0], dout[pr[:, 1]] @ wk[o].T)
    out_row = {tuple(k): r for r, k in enumerate(keys.tolist())}
    got = np.zeros_like(want)
    for (b, z, y, x), src in table.items():
        for o2 in range(27):
            k2 = (o2 // 9, (o2 // 3) % 3, o2 % 3)
            t = (z - 1 + k2[0], y - 1 + k2[1], x - 1 + k2[2])
            if any(v < 0 or v % 2 for v in t):
                continue
            u = tuple(v // 2 for v in t)
            if any(u[a] >= out_shape[a] for a in range(3)) or (b,) + u not in out_row:
                continue
            got[src] += dout[out_row[(b,) + u]] @ wk[26 - o2].T
    rows = sorted(set(table.values()))  # (duplicates: the rows that lost receive nothing in either formulation)
    assert np.abs(got[rows] - want[rows]).max() <= 1e-12 * max(1.0, np.abs(want).max())
    assert np.abs(want[rows]).max() > 0
