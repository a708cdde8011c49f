"""Per-layer precision sensitivity of the decoder MLP — CPU emulation.

For every golden scene the gathered features come from the oracle (fp32); each MLP layer is then evaluated with
its operands rounded the way an MFMA scheme would round them (products exact, fp32 accumulation), the result is
composited with the oracle's raw2outputs and compared with the REFERENCE renderer's rgb_map stored in the fixture.

Schemes (per layer):
    f32      exact fp32
    bf16x3   W_hi.X_hi + W_hi.X_lo + W_lo.X_hi, bf16 parts         (round-1 kernel)
    bf16x1   single bf16 product
    f16x1    single fp16 product
    f16x2w   (W_hi + W_lo).X_hi: weights exact to 22 bits, activations rounded to fp16
    f16x2x   W_hi.(X_hi + X_lo)
    f16x3    three fp16 products
    f16c8    fp16 main product + the two cross terms with BOTH operands rounded to fp8 e4m3 after an exact 2^k block
             scale (what a scaled f8f6f4 MFMA would compute)

    f16c6    fp16 main product + the two cross terms in SIX bits on the scaled f8f6f4 MFMA (which then runs at the rate of
             ONE fp16 K=16 MFMA per K=64, tools/experiments/probe_mxrate.hip): weights fp6 e2m3 with a pack-time E8M0 scale
             per (row, 32 K), activations bf6 e3m2 with a run-time E8M0 scale per (sample, 32 K) from the block's maximum;
             the remainder block reuses the head block's scale - 11

    fold     (fc_0 only, round 4) fc_0 folded into the volume: U_l = fc_0[:, level l] . V_l per voxel in fp32, stored as fp16 head +
             fp16 remainder, interpolated with fp32 weights (the kernel's three fp16 products U_h.Wt_h + U_h.Wt_l + U_l.Wt_h
             drop only the 2^-22 term U_l.Wt_l); fold_h: the head alone (what a single fp16 plane per voxel would give)

Run:  python tools/experiments/precision_sweep.py [--wide | --quick | --levels | --fold]      (CPU, ~1 min)
The same table measured on the HIP kernels is produced by tools/experiments/precision_gpu.py.
"""
import argparse
import itertools
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import neuralbody_oracle as orc  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.golden import scenes  # noqa: E402

LAYERS = ("fc_0", "fc_1", "fc_2", "merged", "view_fc")


def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _f16(x):
    return x.to(torch.float16).to(torch.float32)


def _f8_scaled(x, dim):
    """fp8 e4m3 with one exact power-of-two scale per 32 consecutive elements along `dim` (MX-style block scale)."""
    xm = x.movedim(dim, -1)
    sh = xm.shape
    pad = (-sh[-1]) % 32
    if pad:
        xm = torch.nn.functional.pad(xm, (0, pad))
    blk = xm.reshape(*xm.shape[:-1], -1, 32)
    amax = blk.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - 7.0)  # block max lands in [128, 256) <= 448
    q = (blk / scale).to(torch.float8_e4m3fn).to(torch.float32) * scale
    q = q.reshape(*xm.shape)[..., :sh[-1]]
    return q.movedim(-1, dim)


def _f8_static(x, log2_scale):
    """fp8 e4m3 after a fixed power-of-two pre-scale (saturating at +-448), result scaled back."""
    sc = 2.0 ** log2_scale
    q = (x * sc).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)
    return q / sc


def _bf8_static(x, log2_scale):
    """bf8 e5m2 after a fixed power-of-two pre-scale (saturating at +-57344), result scaled back."""
    sc = 2.0 ** log2_scale
    q = (x * sc).clamp(-57344.0, 57344.0).to(torch.float8_e5m2).to(torch.float32)
    return q / sc


def _minifloat(x, mbits, emin, vmax):
    """round-to-nearest-even onto a sign/exponent/mantissa grid with subnormals below 2^emin, saturating at vmax"""
    a = x.abs()
    e = torch.floor(torch.log2(a.clamp_min(1e-38))).clamp_min(emin)
    step = torch.exp2(e - mbits)
    q = (torch.round(a / step) * step).clamp_max(vmax)
    return torch.sign(x) * q


def _blocks(x, dim):
    xm = x.movedim(dim, -1)
    sh = xm.shape
    pad = (-sh[-1]) % 32
    if pad:
        xm = torch.nn.functional.pad(xm, (0, pad))
    return xm.reshape(*xm.shape[:-1], -1, 32), sh, xm.shape


def _unblocks(q, dim, sh, shp):
    return q.reshape(*shp)[..., :sh[-1]].movedim(-1, dim)


def _fp6_weights(W):
    """fp6 e2m3 (max 7.5), pack-time scale per (row, 32 K): the smallest power of two with max|block| / scale <= 7.5"""
    blk, sh, shp = _blocks(W, 1)
    amax = blk.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    scale = torch.exp2(torch.ceil(torch.log2(amax / 7.5)))
    return _unblocks(_minifloat(blk / scale, 3, 0, 7.5) * scale, 1, sh, shp)


def _fp4_weights(W):
    """fp4 e2m1 (0, 0.5, 1, 1.5, 2, 3, 4, 6), pack-time scale per (row, 32 K): the smallest power of two with max|block| / scale <= 6"""
    blk, sh, shp = _blocks(W, 1)
    amax = blk.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    scale = torch.exp2(torch.ceil(torch.log2(amax / 6.0)))
    return _unblocks(_minifloat(blk / scale, 1, 0, 6.0) * scale, 1, sh, shp)


def _bf6_acts(Xh, Xl):
    """bf6 e3m2 (max 28) of the head and the remainder; run-time scale per (sample, 32 K) = 2^(exponent of the head
    block's maximum - 3) (block max lands in [8, 16)); the remainder block uses that scale * 2^-11"""
    bh, sh, shp = _blocks(Xh, 0)
    bl, _, _ = _blocks(Xl, 0)
    amax = bh.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -100)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - 3.0)
    qh = _minifloat(bh / scale, 2, -2, 28.0) * scale
    sl = scale * 2.0 ** -11
    ql = _minifloat(bl / sl, 2, -2, 28.0) * sl
    return _unblocks(qh, 0, sh, shp), _unblocks(ql, 0, sh, shp)


def _w_scale(W):
    """per-layer static weight scale chosen at pack time: the largest power of two keeping max|W| <= 448"""
    m = float(W.abs().max())
    return 0 if m == 0 else int(np.floor(np.log2(448.0 / m)))


def mm(W, X, scheme):
    """W [M,K] @ X [K,N] with the operand roundings of `scheme` (fp32 accumulation)."""
    if scheme == "f32":
        return W @ X
    if scheme == "bf16x1":
        return _bf16(W) @ _bf16(X)
    if scheme == "f16x1":
        return _f16(W) @ _f16(X)
    if scheme in ("bf16x3", "f16x3", "f16x2w", "f16x2x", "f16c8", "f16c8s", "f16c8b", "f16c6", "f16c6w", "f16c6x", "f16c6-wl", "f16c6-xl", "f16c4", "f16c4h", "f16c4l"):
        r = _bf16 if scheme == "bf16x3" else _f16
        Wh, Xh = r(W), r(X)
        Wl, Xl = r(W - Wh), r(X - Xh)
        if scheme == "f16x2w":
            return Wh @ Xh + Wl @ Xh
        if scheme == "f16x2x":
            return Wh @ Xh + Wh @ Xl
        if scheme == "f16c8":
            return Wh @ Xh + _f8_scaled(Wh, 1) @ _f8_scaled(Xl, 0) + _f8_scaled(Wl, 1) @ _f8_scaled(Xh, 0)
        if scheme == "f16c8s":  # static scales: activations 2^-2 (hi) / 2^8 (lo), weights per layer from max|W|
            return Wh @ Xh + _f8_static(Wh, _w_scale(Wh)) @ _f8_static(Xl, 8) + _f8_static(Wl, _w_scale(Wl)) @ _f8_static(Xh, -2)
        if scheme == "f16c6":
            q_xh, q_xl = _bf6_acts(Xh, Xl)
            return Wh @ Xh + _fp6_weights(Wh) @ q_xl + _fp6_weights(Wl) @ q_xh
        if scheme in ("f16c4", "f16c4h", "f16c4l"):  # round 5: the cross terms' WEIGHT operands in fp4 e2m1 (both / W_h only / W_l only)
            q_xh, q_xl = _bf6_acts(Xh, Xl)
            qh = _fp4_weights(Wh) if scheme != "f16c4l" else _fp6_weights(Wh)
            ql = _fp4_weights(Wl) if scheme != "f16c4h" else _fp6_weights(Wl)
            return Wh @ Xh + qh @ q_xl + ql @ q_xh
        if scheme == "f16c6-wl":  # the W_l cross term dropped (weights rounded to fp16)
            return Wh @ Xh + _fp6_weights(Wh) @ _bf6_acts(Xh, Xl)[1]
        if scheme == "f16c6-xl":  # the X_l cross term dropped (activations rounded to fp16)
            return Wh @ Xh + _fp6_weights(Wl) @ _bf6_acts(Xh, Xl)[0]
        if scheme == "f16c6w":  # only the weights in six bits (activations bf8 as shipped)
            return Wh @ Xh + _fp6_weights(Wh) @ _bf8_static(Xl, 12) + _fp6_weights(Wl) @ _bf8_static(Xh, 0)
        if scheme == "f16c6x":  # only the activations in six bits
            q_xh, q_xl = _bf6_acts(Xh, Xl)
            return Wh @ Xh + _f8_static(Wh, _w_scale(Wh)) @ q_xl + _f8_static(Wl, _w_scale(Wl)) @ q_xh
        if scheme == "f16c8b":  # activations in bf8 e5m2 (no range worries), weights in fp8 e4m3 with pack-time scales
            return Wh @ Xh + _f8_static(Wh, _w_scale(Wh)) @ _bf8_static(Xl, 12) + _f8_static(Wl, _w_scale(Wl)) @ _bf8_static(Xh, 0)
        return Wh @ Xh + Wh @ Xl + Wl @ Xh
    raise ValueError(scheme)


def decode(sd, feat, wpts, viewdir, latent_index, mix, fold=None):
    """raw [N,4] from gathered features [N,352]; `mix` maps layer -> scheme.  Merged feature/latent layer as in
    nb_mlp_pack (fp64 product rounded to fp32, latent folded into the bias)."""
    w = {k: v[..., 0] if v.dim() == 3 else v for k, v in sd.items()}
    x = feat.T
    if mix["fc_0"] in ("fold", "fold_h"):  # round 4: fc_0 . interp(V) = interp(fc_0 . V), the planes as fp16 pairs (or heads only)
        acc, c0 = 0.0, 0
        for vol in fold["vols"]:
            nch = vol.shape[1]
            u = torch.einsum("fc,bcdhw->bfdhw", w["fc_0.weight"][:, c0:c0 + nch], vol)
            uh = _f16(u)
            uq = uh + _f16(u - uh) if mix["fc_0"] == "fold" else uh
            acc = acc + torch.nn.functional.grid_sample(uq, fold["g"], padding_mode="zeros", align_corners=True).reshape(256, -1)
            c0 += nch
        h = torch.relu(acc + w["fc_0.bias"][:, None])
    elif "fc_0_levels" in mix:  # per pyramid level (32 | 64 | 128 | 128 input channels) schemes for fc_0
        acc, c0 = 0.0, 0
        for nch, sch in zip((32, 64, 128, 128), mix["fc_0_levels"]):
            acc = acc + mm(w["fc_0.weight"][:, c0:c0 + nch].contiguous(), x[c0:c0 + nch].contiguous(), sch)
            c0 += nch
        h = torch.relu(acc + w["fc_0.bias"][:, None])
    else:
        h = torch.relu(mm(w["fc_0.weight"], x, mix["fc_0"]) + w["fc_0.bias"][:, None])
    h = torch.relu(mm(w["fc_1.weight"], h, mix["fc_1"]) + w["fc_1.bias"][:, None])
    h = torch.relu(mm(w["fc_2.weight"], h, mix["fc_2"]) + w["fc_2.bias"][:, None])
    alpha = w["alpha_fc.weight"] @ h + w["alpha_fc.bias"][:, None]
    Lw = w["latent_fc.weight"].double()
    Wm = (Lw[:, :256] @ w["feature_fc.weight"].double()).float()
    lat = w["latent.weight"][latent_index].double().reshape(128)
    bm = (Lw[:, :256] @ w["feature_fc.bias"].double() + Lw[:, 256:] @ lat + w["latent_fc.bias"].double()).float()
    pe = torch.cat([orc.embed(viewdir, 4), orc.embed(wpts, 10)], -1).T
    if mix.get("fold"):
        # what nb_march_f16.hip ships since the colour head was folded: view_w[:, :256] . latent_w[:, :256] . feature_w as ONE
        # 128 x 256 layer over fc_2's output (product and bias formed in fp64), plus view_w[:, 256:] over the encodings
        Vg = w["view_fc.weight"][:, :256].double()
        W3 = (Vg @ Wm.double()).float()
        b3 = (Vg @ bm.double() + w["view_fc.bias"].double()).float()
        v = torch.relu(mm(W3, h, mix["view_fc"]) + mm(w["view_fc.weight"][:, 256:].contiguous(), pe, mix.get("view_pe", mix["view_fc"])) + b3[:, None])
    else:
        g = mm(Wm, h, mix["merged"]) + bm[:, None]
        v = torch.relu(mm(w["view_fc.weight"], torch.cat([g, pe], 0), mix["view_fc"]) + w["view_fc.bias"][:, None])
    rgb = w["rgb_fc.weight"] @ v + w["rgb_fc.bias"][:, None]
    return torch.cat([rgb, alpha], 0).T


def scene_inputs(name, widen=None):
    r, sd, body, batch, cam, t_rand = scenes.build(name)
    if widen is not None:
        sd = widen(sd)
    training = r["mode"] == "train"
    sdt, vols, out_sh = H.oracle_volumes(sd, batch, training)
    ray_o, ray_d = torch.from_numpy(batch["ray_o"]), torch.from_numpy(batch["ray_d"])
    near, far = torch.from_numpy(batch["near"]), torch.from_numpy(batch["far"])
    tr = None if t_rand is None else torch.from_numpy(t_rand)
    with torch.no_grad():
        wpts, z = orc.get_sampling_points(ray_o, ray_d, near, far, r["n_samples"], tr)
        vd = ray_d / torch.norm(ray_d, dim=2, keepdim=True)
        ns = r["n_samples"]
        w = wpts.reshape(-1, 3)
        v = vd[:, :, None].repeat(1, 1, ns, 1).reshape(-1, 3)
        pp = orc.pts_to_can_pts(w[None], torch.from_numpy(batch["R"]), torch.from_numpy(batch["Th"]))
        g = orc.get_grid_coords(pp, torch.from_numpy(batch["bounds"]), out_sh, (0.005,) * 3)[:, None, None]
        feat = orc.interpolate_features(g, vols)[0].T.contiguous()
    return dict(r=r, sdt=sdt, feat=feat, w=w, v=v, z=z.reshape(-1, ns), rd=ray_d.reshape(-1, 3),
                li=int(batch["latent_index"][0]), fold=dict(vols=vols, g=g))


def rgb_of(s, mix):
    with torch.no_grad():
        raw = decode(s["sdt"], s["feat"], s["w"], s["v"], s["li"], mix, s["fold"])
        rgb, *_ = orc.raw2outputs(raw.reshape(-1, s["r"]["n_samples"], 4), s["z"], s["rd"], s["r"]["white_bkgd"])
    return rgb.numpy()


def widen_weights(sd):
    """Wide-dynamic-range variant of a state dict: every MLP weight matrix gets per-column and per-row log-uniform
    gains spanning 2^-6..2^6 whose product over a layer pair cancels on average (activations stay O(1))."""
    rs = np.random.RandomState(99)
    sd = dict(sd)
    prev = None
    for name in ("fc_0", "fc_1", "fc_2", "feature_fc", "latent_fc", "view_fc"):
        W = np.array(sd[name + ".weight"])
        out_gain = np.exp2(rs.uniform(-6, 6, W.shape[0])).astype(np.float32)
        if prev is not None and name in ("fc_1", "fc_2", "feature_fc"):
            W[:, :, 0] = W[:, :, 0] / prev[None, :]  # undo the previous layer's row gains (relu is positively homogeneous)
        if name in ("fc_0", "fc_1"):
            W = W * out_gain[:, None, None]
            sd[name + ".bias"] = np.array(sd[name + ".bias"]) * out_gain
            prev = out_gain
        else:
            prev = None
        sd[name + ".weight"] = W.astype(np.float32)
    return sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wide", action="store_true", help="also run the wide-dynamic-range weights variant")
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true", help="only the candidate shipping mixes")
    ap.add_argument("--levels", action="store_true", help="fc_0 sensitivity per pyramid level (one or both cross terms dropped)")
    ap.add_argument("--drop", action="store_true", help="round 4: one cross term of one layer dropped")
    ap.add_argument("--fp4", action="store_true", help="round 5: fp4 e2m1 weight operands in the cross terms (a third less weight stream)")
    ap.add_argument("--fold", action="store_true", help="round 4: fc_0 folded into the volume (fp16 head + remainder planes; heads only)")
    a = ap.parse_args()
    torch.set_num_threads(8)
    names = list(scenes.SCENES)
    data = {n: scene_inputs(n) for n in names}
    gold = {n: H.golden(n)["rgb_map"][0] for n in names}
    if a.wide:
        data["small_wide"] = scene_inputs("small", widen_weights)
        gold["small_wide"] = rgb_of(data["small_wide"], {l: "f32" for l in LAYERS})  # emulation-only reference
    lines = []

    def report(tag, mix):
        errs = {n: float(np.abs(rgb_of(data[n], mix) - gold[n]).max()) for n in data}
        worst = max(errs.values())
        line = "| %-46s | %s | **%.2e** |" % (tag, " | ".join("%.1e" % errs[n] for n in data), worst)
        print(line, flush=True)
        lines.append(line)
        return worst

    print("| mix | " + " | ".join(data) + " | worst |")
    print("|---|" + "---|" * (len(data) + 1))
    if a.fold:
        for f0, tag in (("f16c6", "round 3: gather + fc_0 as f16f6"), ("fold", "SHIPPED round 4: fc_0 folded, fp16 head + remainder planes"),
                        ("fold_h", "fc_0 folded, fp16 heads only"), ("f32", "fc_0 exact")):
            mix = {k: "f16c6" for k in LAYERS}
            mix["fold"] = True
            mix["fc_0"] = f0
            report("%s; fc_1, fc_2, folded colour head f16f6" % tag, mix)
        mix = {k: "f32" for k in LAYERS}
        mix["fc_0"] = "fold"
        report("fc_0 folded (pairs), everything else exact fp32", mix)
        mix["fc_0"] = "fold_h"
        report("fc_0 folded (heads only), everything else exact fp32", mix)
        if a.out:
            with open(a.out, "w") as f:
                f.write("\n".join(lines) + "\n")
        return
    if a.fp4:
        base = {k: "f16c6" for k in LAYERS}
        base.update(fold=True, fc_0="fold", view_pe="f16c6")
        report("shipped round 4 (fp6 e2m3 weights in both cross terms)", base)
        for sc, tag in (("f16c4", "fp4 e2m1 weights in both cross terms"), ("f16c4h", "fp4 for W_h (x X_l) only"), ("f16c4l", "fp4 for W_l (x X_h) only")):
            mix = dict(base)
            for l in ("fc_1", "fc_2", "view_fc", "view_pe"):
                mix[l] = sc
            report("%s, every layer behind fc_0" % tag, mix)
        for layer in ("fc_1", "fc_2", "view_fc", "view_pe"):
            mix = dict(base)
            mix[layer] = "f16c4"
            report("fp4 weights in both cross terms of %s only" % layer, mix)
        mix = dict(base)
        for l in ("fc_1", "fc_2", "view_fc"):
            mix[l] = "f16c4"
        report("fp4 in fc_1, fc_2, colour head over fc_2; encodings fp6", mix)
        if a.out:
            with open(a.out, "w") as f:
                f.write("\n".join(lines) + "\n")
        return
    if a.drop:  # round 4: which cross term of which layer could go (each is 16 of a layer's 96 MFMAs per wave and step)?
        for layer in ("fc_1", "fc_2", "view_fc", "view_pe"):
            for sc in ("f16c6-wl", "f16c6-xl", "f16x1"):
                mix = {k: "f16c6" for k in LAYERS}
                mix["fold"] = True
                mix["fc_0"] = "fold"
                mix["view_pe"] = "f16c6"
                mix[layer] = sc
                report("shipped round 4 except %s = %s" % (layer, sc), mix)
        if a.out:
            with open(a.out, "w") as f:
                f.write("\n".join(lines) + "\n")
        return
    if a.levels:
        for lv in range(4):
            for sch in ("f16x1", "f16x2w", "f16x2x"):
                mix = {k: "f16c6" for k in LAYERS}
                mix["fold"] = True
                lvs = ["f16c6"] * 4
                lvs[lv] = sch
                mix["fc_0_levels"] = lvs
                report("folded f16f6, fc_0 level %d as %s" % (lv, sch), mix)
        return
    if a.quick:
        for pe_s in ("f16x1", "f16x2w", "f16x2x"):
            mix = {k: "f16c6" for k in LAYERS}
            mix["fold"] = True
            mix["view_pe"] = pe_s
            report("folded f16f6, encodings part of view_fc as %s" % pe_s, mix)
        for sc in ("f16c8b", "f16c6"):
            mix = {k: sc for k in LAYERS}
            mix["fold"] = True
            report("SHIPPED (folded colour head) %s" % ("f16f8" if sc == "f16c8b" else "f16f6"), mix)
        report("all bf16x3", {l: "bf16x3" for l in LAYERS})
        report("all f16c8s", {l: "f16c8s" for l in LAYERS})
        mix = {l: "f16c8s" for l in LAYERS}
        mix["merged"] = "f16x1"
        report("f16c8s, merged f16x1", mix)
        mix = {l: "bf16x3" for l in LAYERS}
        mix["merged"] = "f16x1"
        report("bf16x3, merged f16x1", mix)
        report("all f16c8b", {l: "f16c8b" for l in LAYERS})
        mix = {l: "f16c8b" for l in LAYERS}
        mix["merged"] = "f16x1"
        report("f16c8b, merged f16x1", mix)
        for sc in ("f16c6", "f16c6w", "f16c6x"):
            report("all " + sc, {l: sc for l in LAYERS})
            mix = {l: sc for l in LAYERS}
            mix["merged"] = "f16x1"
            report(sc + ", merged f16x1", mix)
        return
    for s in ("f32", "bf16x3", "f16x3", "f16c8", "f16c8b", "f16c6", "f16c6w", "f16c6x", "f16x2w", "f16x2x", "f16x1", "bf16x1"):
        report("all " + s, {l: s for l in LAYERS})
    for s in ("f16x1", "f16c8", "bf16x1"):
        for l in LAYERS:
            mix = {k: "bf16x3" for k in LAYERS}
            mix[l] = s
            report("bf16x3 except %s=%s" % (l, s), mix)
    for sc in ("f16c8b", "f16c6", "f16x1"):  # the shipped arithmetics (colour head folded into one layer), and what a single fp16 product would give there
        mix = {k: ("f16c6" if sc == "f16x1" else sc) for k in LAYERS}
        mix["view_fc"] = sc
        mix["fold"] = True
        report("SHIPPED (folded colour head) %s" % {"f16c8b": "f16f8", "f16c6": "f16f6", "f16x1": "f16f6 trunk, single fp16 product in the folded layer"}[sc], mix)
    for sc in ("f16c8b", "f16c6"):  # before the fold: merged layer as a single fp16 product
        mix = {k: sc for k in LAYERS}
        mix["merged"] = "f16x1"
        report("before the fold, %s: %s, merged f16x1" % ("f16f8" if sc == "f16c8b" else "f16f6", sc), mix)
    for trunk, head in itertools.product(("bf16x3", "f16c8", "f16x1"), ("f16x1", "bf16x1")):
        mix = {"fc_0": trunk, "fc_1": trunk, "fc_2": trunk, "merged": head, "view_fc": head}
        report("trunk %s / colour head %s" % (trunk, head), mix)
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
