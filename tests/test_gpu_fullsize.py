"""Parity AT SIZE: the headline 512x512x64 view and the 1024x1024x128 shape of BASELINE.json configs 3/5,
rendered in train() mode (BatchNorm with batch statistics, as run.py:57,89 and bench.py do), thousands of rays spread over
the image against the oracle marching the same rays through the same feature volumes (the encoder at full out_sh is covered
by test_gpu_parity.py's 'full' scene; re-running its dense stand-in at (96, 352, 192) would take minutes of CPU)."""
import numpy as np
import pytest
import torch

import bench
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _check(size, n_samples, n_check, precision):
    from oracle import neuralbody_oracle as orc

    dev = torch.device(DEV)
    sd, body, net, rend, bd, n = bench.build_scene(dev, size, size, n_samples, precision)
    assert net.training
    pose = bench.build_poses(dev, body, bd, size, size, n_poses=2)[1]
    with torch.no_grad():
        out = rend.render(pose)
        vols = net.encode_sparse_voxels(rend.prepare_sp_input(pose))
    torch.cuda.synchronize()
    assert out["rgb_map"].shape == (1, size * size, 3)
    sel = torch.linspace(0, n - 1, n_check).long()
    b = {k: v.detach().cpu() for k, v in pose.items()}
    b.update(ray_o=b["ray_o"][:, sel], ray_d=b["ray_d"][:, sel], near=b["near"][:, sel], far=b["far"][:, sel])
    with torch.no_grad():
        ref = orc.render(orc.tensor_state_dict(sd), b, n_samples=n_samples, training=True,
                         feature_volume=[v.detach().float().cpu().contiguous() for v in vols])
    # The last sample of a ray has the interval 1e10 (raw2outputs, nerf_net_utils.py:28): its alpha is 0 or 1 by the SIGN of its
    # density.  The default arithmetic recomputes the densities it cannot sign at fp32 level (nb_march's `ill_scratch`), so every
    # ray is held to the SAME tolerance under both arithmetics; only a ray whose last density is within the oracle's own fp32
    # rounding of zero (bench.FP32_SIGMA) is bounded by T_last (the most a flipped last alpha can move) instead.  Rays within
    # bench.ILL_SIGMA of the step are counted and printed: they must be few.
    sigma_last = ref["raw"][0].reshape(n_check, n_samples, 4)[:, -1, 3]
    t_last = (1.0 - ref["weights"][0][:, :-1].sum(1)).numpy()
    ok = (sigma_last.abs() >= bench.ILL_SIGMA)
    assert int((~ok).sum()) <= n_check * bench.ILL_MAX_FRACTION, "too many ill-conditioned rays: %d" % int((~ok).sum())
    okd = ok.to(dev)
    seld = sel.to(dev)[okd]
    okn = ok.numpy()
    all_err = np.abs(out["rgb_map"][0, sel.to(dev)].cpu().numpy() - ref["rgb_map"][0].numpy()).max(1)
    for i in np.nonzero(~okn)[0]:
        bound = H.RGB_TOL if abs(float(sigma_last[i])) > bench.FP32_SIGMA else float(t_last[i]) + H.RGB_TOL
        assert all_err[i] <= bound, "ill-conditioned ray %d: err %.3e > %.3e (sigma_last %.2e, T_last %.2e)" % (
            i, all_err[i], bound, float(sigma_last[i]), t_last[i])
    err = H.assert_close(out["rgb_map"][0, seld].cpu().numpy(), ref["rgb_map"][0].numpy()[okn], H.RGB_TOL, "rgb_map", rel=False)
    H.assert_close(out["acc_map"][0, seld].cpu().numpy(), ref["acc_map"][0].numpy()[okn], 2e-4, "acc_map")
    H.assert_close(out["weights"][0, seld].cpu().numpy(), ref["weights"][0].numpy()[okn], 2e-4, "weights")
    H.assert_close(out["depth_map"][0, seld].cpu().numpy(), ref["depth_map"][0].numpy()[okn], 2e-4, "depth_map")
    assert float(ref["rgb_map"].max() - ref["rgb_map"].min()) > 0.1, "degenerate view"
    print("%dx%dx%d %s: rgb L-inf of %d rays vs oracle %.2e; over all %d rays %.2e (%d ill-conditioned, each inside its flip bound)" % (
        size, size, n_samples, precision, int(ok.sum()), err, n_check, float(all_err.max()), int((~ok).sum())))
    return err


@pytest.mark.parametrize("precision", ["f16f6", "f32"])
def test_headline_view_512x512x64_train_mode(precision):
    _check(512, 64, 4096, precision)


@pytest.mark.parametrize("precision", ["f16f6"])
def test_config3_view_1024x1024x128(precision):
    _check(1024, 128, 1024, precision)


def test_full_size_encoder_matrix_pipe_kernels_match_the_fp32_kernels(monkeypatch):
    """The bench scene's encoder (6.9 k / 22 k / 29 k / 10 k / 1.6 k active rows on levels 0-4) through the fp16-split
    convolution kernels — the per-wave kernel, the LDS-slab kernel, its two-offset-group variant (mid levels) and the
    offset-split kernel (deepest level) — against the exact-fp32 MFMA kernels on the same input: every level of the volume
    pyramid within 2e-5 of its largest value, and the same active set.  (The small fixtures run every layer on the offset-split
    kernel: their row counts are below its 4096-row threshold.)"""
    from neuralbody_amd import network as nbnet

    dev = torch.device(DEV)
    sd, body, net, rend, bd, n = bench.build_scene(dev, 64, 64, 8, "f32")
    sp = rend.prepare_sp_input(bd)
    vols = {}
    for split in (True, False):
        monkeypatch.setattr(nbnet, "ENC_SPLIT", split)
        with torch.no_grad():
            vols[split] = [v.clone() for v in net.encode_sparse_voxels(sp)]
    torch.cuda.synchronize()
    for lvl, (a, b) in enumerate(zip(vols[True], vols[False])):
        scale = float(b.abs().max())
        err = float((a - b).abs().max())
        print("level %d: %s, max |value| %.3f, fp16-split vs fp32 kernels %.3g" % (lvl, tuple(b.shape), scale, err))
        assert scale > 0.1
        assert err <= 2e-5 * scale, (lvl, err, scale)
        assert a.dim() == 5 and a.shape[0] == 1  # [1, C, D, H, W]
        assert torch.equal(a.abs().sum(1) > 0, b.abs().sum(1) > 0)  # the same active voxels


def test_auto_is_a_function_of_the_weights_and_renders_reproducibly():
    """precision='auto' picks its kernel from the packed weights alone (no timing): two Networks with the same
    weights pick the same one, and re-rendering a view gives the same bits."""
    dev = torch.device(DEV)
    sd, body, net, rend, bd, n = bench.build_scene(dev, 512, 512, 64, None)
    assert net.precision == "auto" and net.march_precision() == "f16f6"
    net.eval()  # (train-mode BatchNorm sums its statistics with atomics: the volumes then differ by rounding between two encodes)
    with torch.no_grad():
        first = rend.render(bd)["rgb_map"].clone()
        again = rend.render(bd)["rgb_map"]
    assert torch.equal(first, again)
    sd2, body2, net2, rend2, bd2, _ = bench.build_scene(dev, 64, 64, 8, None)
    assert net2.march_precision() == "f16f6"


def test_full_size_training_gradients_matrix_pipe_vs_fp32_kernels(monkeypatch):
    """One training step's gradients on the bench scene (1024 random rays x 64 jittered samples; 6.9 k - 29 k active rows per
    level) with the encoder's BACKWARD convolutions — backward-input and weight gradient — on the 16-bit matrix pipe (bf16
    pairs, ds_read_b64_tr_b16 fragments) against the same step with those two products on the exact-fp32 MFMA kernels: every
    parameter gradient within 1e-4 of its tensor's largest entry.  The forward is the same in both runs (same ReLU masks: a
    forward that differs in the seventh digit flips a few of the ~10^7 masks and moves gradients by 1e-3, which says nothing
    about the backward kernels); the forward kernels are compared at this size by the test above.  (The fixtures' few hundred
    rows exercise one tile of these kernels.)"""
    from neuralbody_amd import training as nbtrain

    dev = torch.device(DEV)
    grads = {}
    for split in (True, False):
        monkeypatch.setattr(nbtrain, "BWD_INPUT_SPLIT", split)
        sd, body, net, rend, bd, n_rays = bench.build_scene(dev, 512, 512, 64, "f32")
        gen = torch.Generator(device="cpu").manual_seed(0)
        pick = torch.randperm(n_rays, generator=gen)[:1024].to(dev)
        tb = dict(bd)
        for k in ("ray_o", "ray_d", "near", "far"):
            tb[k] = bd[k][:, pick].contiguous()
        tb["mask_at_box"] = torch.ones((1, 1024), dtype=torch.bool, device=dev)
        target = torch.rand((1, 1024, 3), generator=gen).to(dev)
        t_rand = torch.rand((1, 1024, 64), generator=gen).to(dev)
        rend.cfg.perturb = 1.0
        out = rend.render(tb, t_rand=t_rand)
        loss = torch.mean((out["rgb_map"] - target) ** 2)
        loss.backward()
        torch.cuda.synchronize()
        grads[split] = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    assert grads[True].keys() == grads[False].keys() and len(grads[True]) >= 60
    worst = (0.0, None)
    for k, g32 in grads[False].items():
        scale = float(g32.abs().max())
        if scale == 0.0:
            assert float(grads[True][k].abs().max()) == 0.0, k
            continue
        err = float((grads[True][k] - g32).abs().max()) / scale
        worst = max(worst, (err, k))
        assert err <= 1e-4, (k, err)
    print("worst relative gradient difference, matrix pipe vs fp32 kernels: %.2e (%s)" % worst)
