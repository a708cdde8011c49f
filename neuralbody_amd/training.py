"""Backward pass of the decoder (training step).

Forward in training mode decodes the sample points with the activation tap on (`nb_decode_points(dbg=...)`:
F | h1 | h2 | h3 | G | V | PE per point) and composites with `nb_composite`; the backward below consumes the tap:

    d rgb_map --nb_composite_bwd--> d raw --MLP backward (fp32 MFMA GEMMs with fused ReLU-mask / bias-sum epilogues)-->
    parameter gradients + dF --nb_trilinear_bwd--> gradients of the active rows of the four feature volumes

The forward is exact fp32.  The backward's large GEMMs (weight gradients with >= 1024 rows, input gradients with >= 1024 rows
and K % 32 == 0) and the encoder's >= 32-channel backward convolutions run on the 16-bit matrix pipe with both operands as bf16
head + remainder pairs (three products, fp32 accumulation: ~2^-16 relative per product; measured gradient error against float64
autograd 1.2e-5); `NB_BWD_SPLIT=0` / `NB_ENC_SPLIT=0` in the environment keep the decoder GEMMs / the encoder
convolutions on the exact-fp32 kernels.  The merged feature_fc/latent_fc layer of the inference kernels is NOT used here:
gradients are taken layer by layer exactly as the reference modules are written (lib/networks/latent_xyzc.py:99-121).
"""
import os

import torch

from . import ops
from .ops import TAP


def _w2(conv):
    """Conv1d(k=1) weight [out,in,1] -> contiguous [out,in] view."""
    return conv.weight.detach()[:, :, 0]


def decoder_backward(net, tap, d_raw, latent_index, arena=None):
    """Gradients of the MLP parameters and of the gathered features.

    tap [N,1600] activation tap, d_raw [N,4] = d(rgb logits, sigma).  Returns (grads, dF) where grads maps the
    reference's parameter names to gradient tensors and dF is [N,352].  Every product is an fp32 MFMA kernel of
    libnb_hip.so (ops.sgemm); the input-gradient products carry the ReLU mask of the layer below and that layer's bias
    gradient (column sums) in their epilogue."""
    N = tap.shape[0]
    dev = tap.device
    sl = lambda k: tap[:, TAP[k][0]:TAP[k][1]]  # noqa: E731  column slices (row stride 1600)
    F, h1, h2, h3, G, V, PE = (sl(k) for k in ("F", "h1", "h2", "h3", "G", "V", "PE"))
    d_rgb, d_sig = d_raw[:, 0:3], d_raw[:, 3:4]
    W0, W1, W2 = _w2(net.fc_0), _w2(net.fc_1), _w2(net.fc_2)
    Wa, Wf, Wl, Wv, Wr = _w2(net.alpha_fc), _w2(net.feature_fc), _w2(net.latent_fc), _w2(net.view_fc), _w2(net.rgb_fc)
    # accumulators (column sums): views of the backward pass's one zero fill when an arena is given
    zeros = (lambda n: arena.take(n)) if arena is not None else (lambda n: torch.zeros(n, dtype=torch.float32, device=dev))  # noqa: E731
    g = {}
    # rgb_fc (latent_xyzc.py:121)
    g["rgb_fc.weight"] = ops.sgemm(d_rgb, V, trans_a=True)
    g["rgb_fc.bias"] = ops.colsum(d_rgb, out=zeros(3))
    g["alpha_fc.bias"] = ops.colsum(d_sig, out=zeros(1))
    # dV = (d_rgb . W_rgb) * [V > 0]; its column sums are view_fc's bias gradient
    g["view_fc.bias"] = zeros(128)
    dV = ops.sgemm(d_rgb, Wr, relu_mask=V, colsum=g["view_fc.bias"])
    # view_fc on [latent_fc out | PE(viewdir) | PE(xyz)] (:113-120)
    gWv = torch.empty_like(Wv)
    ops.sgemm(dV, G, trans_a=True, out=gWv[:, 0:256])
    ops.sgemm(dV, PE, trans_a=True, out=gWv[:, 256:346])
    g["view_fc.weight"] = gWv
    # dG = dV . W_view[:, :256]; column sums = latent_fc's bias gradient
    sum_dG = zeros(256)
    # (a contiguous copy: rows of the [128, 346] weight are not 16-byte aligned, and the aligned kernels are 4x faster than the
    # general one — one 128 KB copy against 180 us)
    dG = ops.sgemm(dV, Wv[:, 0:256].contiguous(), colsum=sum_dG)
    # latent_fc on [feature_fc out | latent] (:106-111): feature_fc's output is recomputed from h3 (not in the tap)
    latent = net.latent.weight.detach().index_select(0, latent_index.reshape(-1)[:1].long())  # [1,128]
    feat = net.feature_fc.bias.detach()[None].expand(N, 256).contiguous()
    ops.sgemm(h3, Wf, trans_b=True, out=feat, beta=1.0)
    gWl = torch.empty_like(Wl)
    ops.sgemm(dG, feat, trans_a=True, out=gWl[:, 0:256])
    ops.sgemm(sum_dG[:, None], latent, out=gWl[:, 256:384])  # outer product: every sample sees the same latent
    g["latent_fc.weight"] = gWl
    g["latent_fc.bias"] = sum_dG
    g["latent.row"] = ops.sgemm(sum_dG[None], Wl[:, 256:384])[0]  # gradient of latent.weight[latent_index]
    # dfeat = dG . W_latent[:, :256]; column sums = feature_fc's bias gradient
    g["feature_fc.bias"] = zeros(256)
    dfeat = ops.sgemm(dG, Wl[:, 0:256], colsum=g["feature_fc.bias"])
    g["feature_fc.weight"] = ops.sgemm(dfeat, h3, trans_a=True)
    g["alpha_fc.weight"] = ops.sgemm(d_sig, h3, trans_a=True)
    # dh3 = (dfeat . W_feature + d_sigma . W_alpha) * [h3 > 0]; column sums = fc_2's bias gradient
    dh = ops.sgemm(dfeat, Wf)
    g["fc_2.bias"] = zeros(256)
    ops.sgemm(d_sig, Wa, out=dh, beta=1.0, relu_mask=h3, colsum=g["fc_2.bias"])
    # trunk (:99-101): each input-gradient product applies the mask of the layer below and sums its bias gradient
    g["fc_2.weight"] = ops.sgemm(dh, h2, trans_a=True)
    g["fc_1.bias"] = zeros(256)
    dh = ops.sgemm(dh, W2, relu_mask=h2, colsum=g["fc_1.bias"])
    g["fc_1.weight"] = ops.sgemm(dh, h1, trans_a=True)
    g["fc_0.bias"] = zeros(256)
    dh = ops.sgemm(dh, W1, relu_mask=h1, colsum=g["fc_0.bias"])
    g["fc_0.weight"] = ops.sgemm(dh, F, trans_a=True)
    dF = ops.sgemm(dh, W0)
    return g, dF


USE_ARENA = os.environ.get("NB_BWD_ARENA", "1") != "0"  # one zero fill per backward pass instead of one per accumulator
BWD_INPUT_SPLIT = os.environ.get("NB_ENC_SPLIT", "1") != "0"  # backward-input convolutions on the 16-bit matrix pipe (bf16 pairs)
BWD_INPUT_STRIDED = os.environ.get("NB_ENC_SPLIT_STRIDED", "1") != "0"  # ... those of the strided layers too (0: exact-fp32 kernel)


def bwd_input_on_pipe(cin, stride):
    """Does a layer's backward-input product run on the matrix-pipe convolution kernels (which write their result) rather than
    on the exact-fp32 kernel (which scatters with atomics into a zeroed buffer)?  ONE predicate for the arena plan and the pass."""
    return bool(BWD_INPUT_SPLIT and cin >= 32 and (stride == 1 or BWD_INPUT_STRIDED))


DECODER_ARENA = [((n,), torch.float32) for n in (3, 1, 128, 256, 256, 256, 256, 256)]  # decoder_backward's zeros(), in order


def arena_requests(net, ctx):
    """Everything one backward pass accumulates into, in the order it is taken (RenderFunction.backward): the decoder's column
    sums, the gradients of the four levels' active rows, and per encoder layer (last to first) the BatchNorm sums, the weight
    gradient and — where the exact-fp32 backward-input kernel scatters with atomics — the input-row gradient; the vertex-code
    and latent-table gradients."""
    req = list(DECODER_ARENA)
    layers = ctx[1:]
    for rec in layers:
        if rec["level"] is not None:
            req.append(((max(int(rec["n_out_max"]), 1), int(rec["y"].shape[1])), torch.float32))
    for rec in reversed(layers):
        w = rec["conv"].weight
        cin, cout = int(w.shape[3]), int(w.shape[4])
        req.append(((2 * cout,), torch.float64))
        req.append(((3, 3, 3, cin, cout), torch.float32))
        if not bwd_input_on_pipe(cin, rec["stride"]):
            req.append(((max(int(rec["n_in_max"]), 1), cin), torch.float32))
    req.append(((6890, 16), torch.float32))
    req.append((tuple(net.latent.weight.shape), torch.float32))
    return req


def encoder_backward(xyzc_net, ctx, drows_dense, arena=None):
    """Backward of SparseConvNet.forward(save=ctx).  drows_dense[l] [n_rows_l, C_l]: gradient w.r.t. the active rows of
    dense level l (from nb_trilinear_bwd).  Returns (grads, dcodes): grads maps `xyzc_net.<block>.<k>.weight|bias` to
    gradient tensors, dcodes is the gradient of the 6890 x 16 vertex codes."""
    head, layers = ctx[0], ctx[1:]
    names = {}
    for bname, cin, cout, n, stride in _blocks():
        block = getattr(xyzc_net, bname)
        for j in range(n):
            names[id(block[3 * j])] = "xyzc_net.%s.%d" % (bname, 3 * j)
            names[id(block[3 * j + 1])] = "xyzc_net.%s.%d" % (bname, 3 * j + 1)
    g = {}
    dy = None
    rulebooks = {}  # (input grid, output rows, stride) -> [neighbour table]: the layers of a level share it
    for rec in reversed(layers):
        y, x = rec["y"], rec["x"]
        if rec["level"] is not None:
            d = drows_dense[rec["level"]]
            dy = d if dy is None else dy.add_(d)
        if dy is None:
            raise RuntimeError("no gradient reaches the last encoder layer")
        conv, bn = rec["conv"], rec["bn"]
        w = conv.weight.detach()
        cin, cout = int(w.shape[3]), int(w.shape[4])
        # a layer's backward-input product is a convolution of its own — mirrored offsets, transposed slabs; for a strided layer the
        # TRANSPOSED gather (nb_enc_conv16 with stride = -2: input voxel p takes output voxel (p - 1 + k) / 2 where that divides) —
        # and runs on the forward's matrix-pipe kernels with bf16 head / remainder operands (NB_CONV_BF16; gradients span more
        # binades than an un-scaled fp16 head holds).  The 16-channel layers keep the exact-fp32 kernel.
        on_pipe = bwd_input_on_pipe(cin, rec["stride"])
        dx_split = None
        take = (lambda shape, dt=torch.float32: arena.take(shape, dt)) if arena is not None else (lambda shape, dt=torch.float32: None)  # noqa: E731
        if BWD_INPUT_SPLIT and cin >= 32:  # (the weight gradient of every >= 32-channel layer takes the planes too)
            dx, dgamma, dbeta, dx_split = ops.enc_bn_relu_bwd(dy, y, x, rec["n_out"], rec["n_out_max"], rec["bstats"], bn.eps,
                                                              bn.weight.detach(), want_split=True, sums=take((2 * cout,), torch.float64))
        else:
            dx, dgamma, dbeta = ops.enc_bn_relu_bwd(dy, y, x, rec["n_out"], rec["n_out_max"], rec["bstats"], bn.eps,
                                                    bn.weight.detach(), sums=take((2 * cout,), torch.float64))
        g[names[id(bn)] + ".weight"] = dgamma
        g[names[id(bn)] + ".bias"] = dbeta
        g[names[id(conv)] + ".weight"] = ops.enc_conv_bwd_weight(rec["in_rows"], rec["in_grid"], rec["in_dhw"], rec["out_lin"],
                                                               rec["n_out"], rec["n_out_max"], rec["out_dhw"], rec["stride"],
                                                               dx, cin, cout, dx_split=dx_split, out=take((3, 3, 3, cin, cout)),
                                                               rulebook=rulebooks.setdefault(
                                                                   (rec["in_grid"].data_ptr(), rec["out_lin"].data_ptr(), rec["stride"]), []))
        if on_pipe:
            # (the kernel's BatchNorm sums of dIn are of no use: they go to a scratch nobody zeroes or reads — no memset per call)
            dy = ops.enc_conv16(dx_split, rec["out_grid"], rec["out_dhw"], rec["in_lin"], rec["n_in"], rec["n_in_max"],
                                rec["in_dhw"], 1 if rec["stride"] == 1 else -rec["stride"],
                                xyzc_net._packed16(conv, backward_input=True), cout, cin, bf16=True,
                                stats=_sums_sink(dy.device, 2 * cin))[0]
        else:
            dy = ops.enc_conv_bwd_input(dx, rec["out_grid"], rec["out_dhw"], rec["in_lin"], rec["n_in"], rec["n_in_max"],
                                        rec["in_dhw"], rec["stride"], w, out=take((max(int(rec["n_in_max"]), 1), cin)))
    dcodes = ops.enc_scatter_codes_bwd(dy, head["rows_vert"], head["n_rows"], head["n_max"], 6890,
                                       out=arena.take((6890, 16)) if arena is not None else None)
    return g, dcodes


_SINKS = {}


def _sums_sink(device, n):
    """fp64 [n] scratch for statistics outputs that are never read (one per device and size)."""
    key = (str(device), n)
    if key not in _SINKS:
        _SINKS[key] = torch.empty(n, dtype=torch.float64, device=device)
    return _SINKS[key]


def _blocks():
    from .network import ENCODER_BLOCKS

    return ENCODER_BLOCKS


class RenderFunction(torch.autograd.Function):
    """Differentiable Renderer.render for the training step (lib/train/trainers/if_nerf_clight.py:18-36): forward and
    backward are both the HIP path; autograd only carries d rgb_map in and the parameter gradients out."""

    @staticmethod
    def forward(ctx, renderer, batch, t_rand, raw_noise, *params):
        net, cfg = renderer.net, renderer.cfg
        if not net.training:
            # nb_enc_bn_relu_bwd implements the batch-statistics BatchNorm backward only; in eval() the forward normalises
            # with the running statistics, for which that formula is wrong
            raise RuntimeError("the differentiable HIP path needs net.train() (BatchNorm with batch statistics, as the "
                               "reference trains: lib/train/trainers/trainer.py:28); use torch.no_grad() for eval-mode renders")
        sp_input = renderer.prepare_sp_input(batch)
        enc_ctx = []
        with torch.no_grad():
            vols = net.encode_sparse_voxels(sp_input, save=enc_ctx)
            ray_o, ray_d, near, far = batch["ray_o"], batch["ray_d"], batch["near"], batch["far"]
            wpts, z_vals = renderer.get_sampling_points(ray_o, ray_d, near, far, t_rand)
            n_batch, n_pixel, S = wpts.shape[:3]
            viewdir = ray_d / torch.norm(ray_d, dim=2, keepdim=True)
            w = wpts.reshape(-1, 3).float().contiguous()
            v = viewdir[:, :, None].expand(n_batch, n_pixel, S, 3).reshape(-1, 3).float().contiguous()
            scene = net.make_scene(vols, sp_input)
            lb = net.latent_bias(sp_input["latent_index"])
            raw, tap = ops.decode_points(scene, net.packed_weights("f32"), lb, w, v, debug=True, precision="f32")
            if raw_noise is not None:
                # raw2outputs adds randn * raw_noise_std to the densities in front of the relu (nerf_net_utils.py:31-35): the noisy
                # densities are what is composited AND what the backward differentiates through (d sigma is unchanged by an addend)
                raw = raw.clone()
                raw.view(-1, S, 4)[..., 3] += raw_noise.reshape(-1, S).to(raw) * float(cfg.raw_noise_std)
            z = z_vals.reshape(-1, S).float().contiguous()
            rd = ray_d.reshape(-1, 3).float().contiguous()
            rgb, disp, acc, weights, depth = ops.composite(raw.view(-1, S, 4), z, rd, cfg.white_bkgd)
        if DEBUG_CAPTURE is not None:
            DEBUG_CAPTURE.update(tap=tap, enc_ctx=enc_ctx)
        ctx.renderer, ctx.sp_input, ctx.enc_ctx, ctx.scene = renderer, sp_input, enc_ctx, scene
        ctx.saved = (raw, tap, z, rd, w)
        ctx.n_params = len(params)
        ctx.mark_non_differentiable(disp, acc, weights, depth)
        return (rgb.view(n_batch, n_pixel, 3), disp.view(n_batch, n_pixel), acc.view(n_batch, n_pixel),
                weights.view(n_batch, n_pixel, S), depth.view(n_batch, n_pixel))

    @staticmethod
    def backward(ctx, d_rgb, *_unused):
        renderer = ctx.renderer
        net, cfg = renderer.net, renderer.cfg
        raw, tap, z, rd, w = ctx.saved
        S = z.shape[1]
        d_raw = ops.composite_bwd(raw.view(-1, S, 4), z, rd, d_rgb.reshape(-1, 3).float().contiguous(), cfg.white_bkgd)
        li = ctx.sp_input["latent_index"]
        # ONE zero fill for every accumulator of the pass (column sums, atomics' targets: ~60 fills and memsets otherwise)
        arena = ops.ZeroArena(ops.ZeroArena.size_of(arena_requests(net, ctx.enc_ctx)), w.device) if USE_ARENA else None
        g, dF = decoder_backward(net, tap, d_raw.view(-1, 4), li, arena)
        layers = ctx.enc_ctx[1:]
        dense = [rec for rec in layers if rec["level"] is not None]
        grids = [rec["out_grid"] for rec in dense]
        if arena is not None:
            drows = [arena.take((max(int(rec["n_out_max"]), 1), rec["y"].shape[1])) for rec in dense]
        else:
            drows = [torch.zeros((rec["n_out_max"] if rec["n_out_max"] > 0 else 1, rec["y"].shape[1]), dtype=torch.float32,
                                 device=w.device) for rec in dense]
        ops.trilinear_bwd(ctx.scene, grids, drows, w, dF, run_length=S)  # w: [rays x S, 3], a ray's samples consecutive
        ge, dcodes = encoder_backward(net.xyzc_net, ctx.enc_ctx, drows, arena)
        g.update(ge)
        g["c.weight"] = dcodes
        glat = arena.take(tuple(net.latent.weight.shape)) if arena is not None else torch.zeros_like(net.latent.weight)
        if arena is not None and arena.off != arena.buf.numel() - arena.slack:
            raise RuntimeError("backward pass took %d bytes of its zero arena, the plan (arena_requests) holds %d: the plan no longer "
                               "mirrors the pass" % (arena.off, arena.buf.numel() - arena.slack))
        glat.index_copy_(0, li.reshape(-1)[:1].long(), g.pop("latent.row")[None])
        g["latent.weight"] = glat
        out = []
        for name, p in net.named_parameters():
            gr = g.get(name)
            out.append(None if gr is None else gr.reshape(p.shape))
        assert len(out) == ctx.n_params
        return (None, None, None, None) + tuple(out)


# tests/test_gpu_backward.py sets this to a dict to receive the forward's activations (MLP tap, encoder records) so that
# its float64 reference can differentiate through the SAME ReLU masks
DEBUG_CAPTURE = None
MAX_TAP_BYTES = 32 * 2 ** 30  # activation tap kept for the backward pass (ops.DBG_WIDTH floats per sample)


def render_train(renderer, batch, t_rand=None, raw_noise=None):
    """Renderer.render with gradients: same output dict, rgb_map carries the autograd graph.  raw_noise: standard-normal
    [B, n_rays, N_samples] used when cfg.raw_noise_std != 0 (drawn here when None)."""
    n_points = batch["ray_o"].shape[1] * renderer.cfg.N_samples
    tap_bytes = n_points * ops.DBG_WIDTH * 4
    if tap_bytes > MAX_TAP_BYTES:
        raise RuntimeError(
            "differentiable render of %d rays x %d samples would keep %.1f GiB of activations for the backward pass "
            "(limit %.0f GiB, neuralbody_amd.training.MAX_TAP_BYTES). Training batches are cfg.N_rand = 1024 rays "
            "(if_nerf_clight.py); full-image inference must run under torch.no_grad() as run.py:66,98 does."
            % (batch["ray_o"].shape[1], renderer.cfg.N_samples, tap_bytes / 2 ** 30, MAX_TAP_BYTES / 2 ** 30))
    params = [p for _, p in renderer.net.named_parameters()]
    if float(renderer.cfg.raw_noise_std) != 0.0:
        if raw_noise is None:
            raw_noise = torch.randn((batch["ray_o"].shape[0], batch["ray_o"].shape[1], renderer.cfg.N_samples), device=batch["ray_o"].device)
    else:
        raw_noise = None
    rgb, disp, acc, weights, depth = RenderFunction.apply(renderer, batch, t_rand, raw_noise, *params)
    return {"rgb_map": rgb, "disp_map": disp, "acc_map": acc, "weights": weights, "depth_map": depth}
