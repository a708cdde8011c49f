"""Backward pass of the decoder (training step).

Forward in training mode decodes the sample points with the activation tap on (`nb_decode_points(dbg=...)`:
F | h1 | h2 | h3 | G | V | PE per point) and composites with `nb_composite`; the backward below consumes the tap:

    d rgb_map --nb_composite_bwd--> d raw --MLP backward (rocBLAS GEMMs via nb_sgemm, nb_relu_bwd, nb_colsum)-->
    parameter gradients + dF --nb_trilinear_bwd--> gradients of the active rows of the four feature volumes

All arithmetic is fp32.  The merged feature_fc/latent_fc layer of the inference kernels is NOT used here: gradients are
taken layer by layer exactly as the reference modules are written (lib/networks/latent_xyzc.py:99-121).
"""
import torch

from . import ops
from .ops import TAP


def _w2(conv):
    """Conv1d(k=1) weight [out,in,1] -> contiguous [out,in] view."""
    return conv.weight.detach()[:, :, 0]


def decoder_backward(net, tap, d_raw, latent_index):
    """Gradients of the MLP parameters and of the gathered features.

    tap [N,1600] activation tap, d_raw [N,4] = d(rgb logits, sigma).  Returns (grads, dF) where grads maps the
    reference's parameter names to gradient tensors and dF is [N,352]."""
    N = tap.shape[0]
    sl = lambda k: tap[:, TAP[k][0]:TAP[k][1]]  # noqa: E731  column slices (row stride 1600)
    F, h1, h2, h3, G, V, PE = (sl(k) for k in ("F", "h1", "h2", "h3", "G", "V", "PE"))
    d_rgb, d_sig = d_raw[:, 0:3], d_raw[:, 3:4]
    W0, W1, W2 = _w2(net.fc_0), _w2(net.fc_1), _w2(net.fc_2)
    Wa, Wf, Wl, Wv, Wr = _w2(net.alpha_fc), _w2(net.feature_fc), _w2(net.latent_fc), _w2(net.view_fc), _w2(net.rgb_fc)
    g = {}
    # rgb_fc (latent_xyzc.py:121)
    g["rgb_fc.weight"] = ops.sgemm(d_rgb, V, trans_a=True)
    g["rgb_fc.bias"] = ops.colsum(d_rgb)
    dV = ops.sgemm(d_rgb, Wr)
    ops.relu_bwd_(dV, V.contiguous())
    # view_fc on [latent_fc out | PE(viewdir) | PE(xyz)] (:113-120)
    gWv = torch.empty_like(Wv)
    ops.sgemm(dV, G, trans_a=True, out=gWv[:, 0:256])
    ops.sgemm(dV, PE, trans_a=True, out=gWv[:, 256:346])
    g["view_fc.weight"] = gWv
    g["view_fc.bias"] = ops.colsum(dV)
    dG = ops.sgemm(dV, Wv[:, 0:256])
    # latent_fc on [feature_fc out | latent] (:106-111)
    latent = net.latent.weight.detach().index_select(0, latent_index.reshape(-1)[:1].long())  # [1,128]
    feat = net.feature_fc.bias.detach()[None].expand(N, 256).contiguous()
    ops.sgemm(h3, Wf, trans_b=True, out=feat, beta=1.0)
    sum_dG = ops.colsum(dG)
    gWl = torch.empty_like(Wl)
    ops.sgemm(dG, feat, trans_a=True, out=gWl[:, 0:256])
    ops.sgemm(sum_dG[:, None], latent, out=gWl[:, 256:384])  # outer product: every sample sees the same latent
    g["latent_fc.weight"] = gWl
    g["latent_fc.bias"] = sum_dG
    g["latent.row"] = ops.sgemm(sum_dG[None], Wl[:, 256:384])[0]  # gradient of latent.weight[latent_index]
    dfeat = ops.sgemm(dG, Wl[:, 0:256])
    # feature_fc, alpha_fc (:103, :106)
    g["feature_fc.weight"] = ops.sgemm(dfeat, h3, trans_a=True)
    g["feature_fc.bias"] = ops.colsum(dfeat)
    g["alpha_fc.weight"] = ops.sgemm(d_sig, h3, trans_a=True)
    g["alpha_fc.bias"] = ops.colsum(d_sig)
    dh = ops.sgemm(dfeat, Wf)
    ops.sgemm(d_sig, Wa, out=dh, beta=1.0)
    # trunk (:99-101)
    for name, W, h_out, h_in in (("fc_2", W2, h3, h2), ("fc_1", W1, h2, h1), ("fc_0", W0, h1, F)):
        ops.relu_bwd_(dh, h_out.contiguous())
        g[name + ".weight"] = ops.sgemm(dh, h_in, trans_a=True)
        g[name + ".bias"] = ops.colsum(dh)
        dh = ops.sgemm(dh, W)
    return g, dh  # dh is now dF [N,352]
