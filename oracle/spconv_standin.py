"""TEST INFRASTRUCTURE ONLY — dense-equivalent restatement of the spconv 1.x API
surface that the reference's encoder uses.  Never imported by the product path.

The reference depends on spconv v1.2.1 @ abf0acf30f5526ea93e687e3f424f62d9cd8313a
(/root/reference INSTALL.md:15-22, docker/spconv.sh:1-5), a C++/CUDA library
whose source is NOT under /root/reference and cannot be built offline.
**Parity with spconv itself is therefore UNPINNED** (SURVEY.md §8(c), §A.3);
this module restates its published semantics:

  * SparseConvTensor(features[N,C], indices[N,4]=(b,z,y,x), spatial_shape, batch_size)
        call site: lib/networks/latent_xyzc.py:36,137
  * SubMConv3d(Cin,Cout,3,bias=False): output active set == input active set,
        out[p] = sum_o W[o] . in[p+o-1] over ACTIVE neighbours (cross-correlation)
        == conv3d(pad=1) on the zero-filled grid, masked to the active set.
        call sites: lib/networks/latent_xyzc.py:220-258
  * SparseConv3d(Cin,Cout,3,2,padding=1,bias=False): output active where >=1 active
        input lies in the 3^3 receptive field (== max_pool3d(mask,3,2,1)>0); values =
        conv3d(stride=2,pad=1) masked to that set.  call site: :265-274
  * weight parameter layout [kD,kH,kW,Cin,Cout]
  * SparseSequential applies BatchNorm1d/ReLU to the [N_active, C] row matrix, so
        BN statistics are over ACTIVE rows only.
  * .dense() -> [B,C,D,H,W] zeros at inactive sites.  call sites: :189,193,197,201

Deterministic duplicate rule (spconv leaves duplicates implementation-defined):
the LAST row that maps to a voxel wins at scatter; BN counts unique voxels.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, mask=None):
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        if grid is not None:
            self.grid, self.mask = grid, mask
            return
        D, H, W = self.spatial_shape
        C = features.shape[1]
        idx = indices.long()
        lin = ((idx[:, 0] * D + idx[:, 1]) * H + idx[:, 2]) * W + idx[:, 3]
        # last row wins, deterministically (works with autograd: gather of the winning row)
        n_cells = self.batch_size * D * H * W
        winner = torch.full((n_cells,), -1, dtype=torch.long, device=features.device)
        order = torch.arange(lin.numel(), device=features.device)
        winner.scatter_reduce_(0, lin, order, reduce="amax", include_self=True)
        active = winner >= 0
        flat = torch.zeros(n_cells, C, dtype=features.dtype, device=features.device)
        flat[active] = features[winner[active]]
        self.grid = flat.view(self.batch_size, D, H, W, C).permute(0, 4, 1, 2, 3).contiguous()
        self.mask = active.view(self.batch_size, 1, D, H, W)

    def dense(self):
        return self.grid

    # row views used by SparseSequential for BN / ReLU
    def rows(self):
        m = self.mask[:, 0]
        return self.grid.permute(0, 2, 3, 4, 1)[m]  # [N_active, C]

    def with_rows(self, rows):
        m = self.mask[:, 0]
        B, _, D, H, W = self.grid.shape
        g = torch.zeros(B, D, H, W, rows.shape[1], dtype=rows.dtype, device=rows.device)
        g[m] = rows
        return SparseConvTensor(None, None, self.spatial_shape, self.batch_size,
                                grid=g.permute(0, 4, 1, 2, 3).contiguous(), mask=self.mask)


class _SpConvBase(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, indice_key=None):
        super().__init__()
        assert not bias
        k = kernel_size
        self.kernel_size, self.stride, self.padding = k, stride, padding
        self.in_channels, self.out_channels = in_channels, out_channels
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.randn(k, k, k, in_channels, out_channels) / (k * k * k * in_channels) ** 0.5)

    def _w(self):
        return self.weight.permute(4, 3, 0, 1, 2)


class SubMConv3d(_SpConvBase):
    def forward(self, x):
        k = self.kernel_size
        out = F.conv3d(x.grid, self._w(), padding=k // 2) * x.mask.to(x.grid.dtype)
        return SparseConvTensor(None, None, x.spatial_shape, x.batch_size, grid=out, mask=x.mask)


class SparseConv3d(_SpConvBase):
    def forward(self, x):
        k, s, p = self.kernel_size, self.stride, self.padding
        mask = F.max_pool3d(x.mask.to(x.grid.dtype), k, s, p) > 0
        out = F.conv3d(x.grid, self._w(), stride=s, padding=p) * mask.to(x.grid.dtype)
        return SparseConvTensor(None, None, list(out.shape[2:]), x.batch_size, grid=out, mask=mask)


class SparseSequential(nn.Sequential):
    def forward(self, x):
        for m in self:
            if isinstance(m, _SpConvBase):
                x = m(x)
            else:
                x = x.with_rows(m(x.rows()))
        return x
