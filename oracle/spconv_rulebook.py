"""TEST INFRASTRUCTURE ONLY — a SECOND, independent restatement of the two spconv 1.x layers the reference's encoder uses
(lib/networks/latent_xyzc.py:208-274), written the way spconv itself computes them: hash table of active voxels -> rulebook of
(input row, output row) pairs per kernel offset -> gather, GEMM with that offset's [Cin, Cout] slab, scatter-add.  Never imported by
the product path.

spconv v1.2.1 @ abf0acf30f5526ea93e687e3f424f62d9cd8313a (INSTALL.md:15-22) is absent and cannot be built offline: **parity with
spconv itself stays UNPINNED**.  This module cannot change that; it removes one risk — that `oracle/spconv_standin.py` (dense
masked conv3d, the formulation every fixture runs on) mis-states the published semantics in a way a differently built
formulation would expose.  tests/test_spconv_restatements.py holds the two against each other on duplicates, borders, odd sizes
under stride 2 and the weight layout.  Semantics restated (spconv 1.x documentation / source layout):

  * indices [N, 4] = (batch, z, y, x); weight [kD, kH, kW, Cin, Cout]; no bias in the reference's layers;
  * SubMConv3d(k=3): output active set = input active set; out[p] = sum over offsets o in {0,1,2}^3 of in[p + o - 1] . W[o] for the
    ACTIVE neighbours p + o - 1 (cross-correlation, no kernel flip);
  * SparseConv3d(k=3, stride=2, padding=1): output size (in + 2 p - k) // s + 1 per axis; output voxel q is active iff some active
    input p and offset o satisfy p + pad - o = s q; out[q] = sum of in[p] . W[o] over those pairs;
  * duplicates in `indices` are implementation-defined in spconv; both restatements use "the last row wins".
"""
import numpy as np


def _hash_rows(indices):
    """voxel (b, z, y, x) -> row, the LAST row of a duplicated voxel winning; returns (table, kept rows in first-seen order)."""
    table = {}
    for r, key in enumerate(map(tuple, np.asarray(indices, dtype=np.int64))):
        table[key] = r
    return table


def subm_rulebook(indices, spatial_shape, k=3):
    """-> (out_keys [M, 4], pairs: list over the k^3 offsets of int64 [n_o, 2] = (input FEATURE row, output row))."""
    table = _hash_rows(indices)
    keys = list(table.keys())  # unique voxels; the output set of a submanifold convolution
    row_of = {key: i for i, key in enumerate(keys)}
    D, H, W = spatial_shape
    pairs = []
    for oz in range(k):
        for oy in range(k):
            for ox in range(k):
                pr = []
                for key, q in row_of.items():
                    b, z, y, x = key
                    nz, ny, nx = z + oz - k // 2, y + oy - k // 2, x + ox - k // 2
                    if 0 <= nz < D and 0 <= ny < H and 0 <= nx < W:
                        src = table.get((b, nz, ny, nx))
                        if src is not None:
                            pr.append((src, q))
                pairs.append(np.asarray(pr, dtype=np.int64).reshape(-1, 2))
    return np.asarray(keys, dtype=np.int64).reshape(-1, 4), pairs


def sparse_rulebook(indices, spatial_shape, k=3, stride=2, padding=1):
    """-> (out_keys [M, 4], out_shape, pairs) of SparseConv3d."""
    table = _hash_rows(indices)
    out_shape = [(s + 2 * padding - k) // stride + 1 for s in spatial_shape]
    out_row = {}
    pairs = [[] for _ in range(k ** 3)]
    for key, src in table.items():
        b, z, y, x = key
        for oz in range(k):
            for oy in range(k):
                for ox in range(k):
                    tz, ty, tx = z + padding - oz, y + padding - oy, x + padding - ox
                    if tz % stride or ty % stride or tx % stride:
                        continue
                    qz, qy, qx = tz // stride, ty // stride, tx // stride
                    if not (0 <= qz < out_shape[0] and 0 <= qy < out_shape[1] and 0 <= qx < out_shape[2]):
                        continue
                    q = out_row.setdefault((b, qz, qy, qx), len(out_row))
                    pairs[(oz * k + oy) * k + ox].append((src, q))
    keys = np.asarray(list(out_row.keys()), dtype=np.int64).reshape(-1, 4)
    return keys, out_shape, [np.asarray(p, dtype=np.int64).reshape(-1, 2) for p in pairs]


def apply_rulebook(features, weight, n_out, pairs):
    """gather -> GEMM -> scatter-add, one kernel offset at a time.  features [N, Cin] float64/32, weight [k, k, k, Cin, Cout]."""
    k = weight.shape[0]
    w = np.asarray(weight).reshape(k ** 3, weight.shape[3], weight.shape[4])
    out = np.zeros((n_out, weight.shape[4]), dtype=np.result_type(features, weight))
    for o, pr in enumerate(pairs):
        if len(pr):
            np.add.at(out, pr[:, 1], np.asarray(features)[pr[:, 0]] @ w[o])
    return out


def dense(keys, rows, batch_size, spatial_shape):
    """.dense(): [B, C, D, H, W], zeros at inactive voxels."""
    D, H, W = spatial_shape
    g = np.zeros((batch_size, rows.shape[1], D, H, W), dtype=rows.dtype)
    if len(keys):
        g[keys[:, 0], :, keys[:, 1], keys[:, 2], keys[:, 3]] = rows
    return g


def batchnorm_relu_rows(rows, gamma, beta, eps=1e-3):
    """BatchNorm1d in training mode over the ACTIVE rows (biased variance), then ReLU — what spconv.SparseSequential applies to the
    [N_active, C] feature matrix (latent_xyzc.py:215)."""
    mean = rows.mean(0)
    var = rows.var(0)
    return np.maximum((rows - mean) / np.sqrt(var + eps) * gamma + beta, 0.0)
