"""bench.py — Neural Body hot-path throughput on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Run it from the repository checkout: the synthetic scenes are test infrastructure (tests/synthetic.py, imported as
`tests.synthetic`), as is the CPU oracle the parity and cpu_baseline legs call (oracle/).  `python bench.py --gpus N` with N > 1 and
no launcher around it re-executes itself under torch.distributed.run (self_launch); an N > 1 weak run also carries a strong leg
(`strong_leg`: one view per step, its rays split over the ranks) behind the timed region.

Workload (BASELINE.json `metric`): synthetic 6890-vertex SMPL scene, 512x512 image whose every pixel
hits the SMPL bounding box, 64 samples/ray, random latents/MLP weights (tests/synthetic.py,
seed 0).  One STEP = `Renderer.render(batch)` for one view per GPU: structured-latent-code encoder
(17 sparse conv+BN+ReLU layers) + fused march (sampling, trilinear gather, MLP, compositing) of
262 144 rays, inputs (rays, vertices, weights) resident in HBM.  With N GPUs the job is N views
(weak scaling): rank r marches view r, then the rendered RGB tiles are all-gathered over RCCL.

Hygiene: the timed region cycles through N_POSES = 8 distinct camera poses whose ray tensors were
generated on the device beforehand (no per-view cache can hit: each step sees other ray / mask tensors), `ms_per_step` is
total / K as the contract says and `median_ms_per_step` is the median of per-step HIP-event times; `parity_linf_all` is the
rgb L-inf of 4096 rays of a timed view against the oracle on the same feature volumes (rank 0, N = 1) — EVERY checked ray, the
figure the pass rule holds to 1e-4 (parity_check(): the only exception is a ray whose last density the oracle's own fp32 rounding
cannot sign, |sigma_last| < FP32_SIGMA, bounded by its T_last); `parity_linf` is the same without the rays whose last sample's
density is within ILL_SIGMA of zero, which `parity` lists one by one together with what the march's last-sample fix-up did.
`--scaling strong` shards ONE view's rays over the ranks (parallel.render_sharded) instead of one view per rank.

The JSON line also carries
  roofline     — the dominant kernel (nb_march_fold_kernel by default; --precision f32 picks the exact one): algorithmic MLP
                 flops (859 904 per ray-sample) / its average launch duration measured with HIP
                 events inside the timed region, against the dense MFMA peak of the arithmetic its main product runs on
                 (2.5 PFLOP/s for the fp16 / bf16 paths, 157.3 TFLOP/s for exact fp32); `executed_frac` = the MFMA work the
                 kernel actually issues, in fp16-equivalent pipe time, over the same peak.
  cpu_baseline — the CPU restatement of the reference (oracle/, torch CPU, all host cores) marching a
                 bounded sample of the same rays through the same feature volumes.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 859904.0  # MLP MACs x 2 as the reference layers are written
# per precision: (dtype tag, kernel, executed MFMA flop/sample, dense MFMA peak TFLOP/s from MI355X_MICROARCH.md)
PRECISION_INFO = {
    # feature_fc.latent_fc merged, latent folded into a bias: 331 648 MAC/sample on v_mfma_f32_32x32x2_f32
    "f32": ("f32", "nb_march_kernel", 663296.0, 157.3),
    # fc_0 folded into the volume; fc_1 / fc_2 / the folded colour head: fp16 main product + two narrow cross terms (a K=64
    # fp4 x bf6 MFMA occupies the matrix pipe as long as ONE K=16 fp16 MFMA, profiles/r05_probe_fp4.log, and is counted as
    # one).  Per wave and depth step of 64 samples: fc_1 96, fc_2 96, the colour head 48 + 24 over the encodings, and 12 per 16
    # voxels of the step's voxel list (8 x 8 pixel tiles of the bench view: 59 voxels = 4.2 chunks on average)
    "f16f6": ("f16+f6/f4", "nb_march_fold_kernel", 4 * (264 + 12 * 4.2) * 32768 / 64.0, 2500.0),
}


def build_scene(dev, H=512, W=512, n_samples=64, precision=None):
    from neuralbody_amd import ops
    from tests import synthetic as syn
    from neuralbody_amd.network import Network
    from neuralbody_amd.renderer import RenderConfig, Renderer

    sd = syn.make_weights(0, num_train_frame=230)
    body = syn.make_body(seed=0)
    K, R, T = syn.full_coverage_camera(body, H, W)
    net = Network(num_train_frame=230, precision=precision)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    net = net.to(dev)
    net.train()  # run.py:57,89 renders in train() mode: BatchNorm uses batch statistics
    ro, rd, near, far, mask, n = ops.raygen(H, W, K, R, T, body["can_bounds"], dev)
    n = int(n)
    assert n == H * W, "the throughput camera must see the bbox in every pixel (%d of %d)" % (n, H * W)
    batch = syn.make_batch(body, np.zeros((1, 3), np.float32), np.zeros((1, 3), np.float32), np.zeros(1, np.float32),
                           np.zeros(1, np.float32), np.ones(1, bool))
    bd = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in batch.items()
          if k not in ("ray_o", "ray_d", "near", "far", "mask_at_box")}
    bd.update(ray_o=ro[None, :n], ray_d=rd[None, :n], near=near[None, :n], far=far[None, :n],
              mask_at_box=mask[None].bool())
    rend = Renderer(net, RenderConfig(N_samples=n_samples, perturb=0.0, H=H, W=W))
    return sd, body, net, rend, bd, n


N_POSES = 8


def build_poses(dev, body, bd, H, W, n_poses=N_POSES):
    """`n_poses` batches of the same frame seen from different full-coverage cameras (yaw steps around the body): rays
    generated on the device by nb_raygen, everything resident in HBM before the timed region."""
    from neuralbody_amd import ops
    from tests import synthetic as syn

    poses = []
    for i in range(n_poses):
        K, R, T = syn.full_coverage_camera(body, H, W, yaw=0.35 + 0.12 * i, pitch=0.1 - 0.03 * i)
        ro, rd, near, far, mask, n = ops.raygen(H, W, K, R, T, body["can_bounds"], dev)
        n = int(n)
        assert n == H * W, "throughput cameras must see the bbox in every pixel (pose %d: %d of %d)" % (i, n, H * W)
        b = {k: v for k, v in bd.items() if k not in ("ray_o", "ray_d", "near", "far", "mask_at_box")}
        b.update(ray_o=ro[None, :n], ray_d=rd[None, :n], near=near[None, :n], far=far[None, :n], mask_at_box=mask[None].bool())
        poses.append(b)
    return poses


# The last sample of a ray has the interval 1e10 (raw2outputs, nerf_net_utils.py:28): alpha_last is 0 or 1 by the SIGN of its
# density, so a ray whose last density is within an arithmetic's own density error of zero can take the other branch: its colour
# then moves by up to T_last (the transmittance that reaches the last sample).  Since round 6 the default arithmetic recomputes
# exactly those densities at fp32 level (nb_march's `ill_scratch`, include/nb_hip.h), so EVERY ray is held to the budget; what
# remains undecidable is the oracle's own fp32 rounding: |sigma_last| < FP32_SIGMA, where two fp32 evaluations of the reference's
# formula may disagree on the sign — such a ray (none in 262 144 on the bench view) is bounded by its T_last and reported.
# ILL_SIGMA (the measured density error of the default arithmetic, ~3e-4, tools/experiments/last_sample_probe.py, with a margin)
# is kept for reporting: `n_ill` / `ill` list the rays inside it with the error actually measured.
FP32_SIGMA = 2e-5
ILL_SIGMA = 5e-4
ILL_MAX_FRACTION = 0.005


def parity_check(sd, net, rend, batch, n_samples, n_check=4096, chunk=16384, list_ill=16):
    """rgb error of `n_check` rays spread over the view (None: EVERY ray): HIP render of the FULL view vs the oracle marching
    the picked rays through the same feature volumes (train-mode BatchNorm like the timed region), `chunk` rays at a time.
    Returns a dict:
      linf_all        max over ALL checked rays (no exclusion)
      linf            max over the well-conditioned rays (|sigma_last| >= ILL_SIGMA in the oracle), budget 1e-4 (north_star)
      n_flipped       ill-conditioned rays beyond 1e-4 (their last alpha took the other branch; bounded by T_last each)
      n, n_ill        how many rays each of the two sets holds
      ill             per ill-conditioned checked ray (the first `list_ill`): oracle sigma_last, T_last (transmittance in front of
                      the last sample: the most a flipped alpha_last could move), the rgb error actually measured
      ill_linf        the largest error among the ill-conditioned rays
      ill_full_view   the same criterion counted over EVERY ray of the view (from the HIP path's own raw output)."""
    from oracle import neuralbody_oracle as orc

    with torch.no_grad():
        out = rend.render(batch, want_raw=True)
        vols = net.encode_sparse_voxels(rend.prepare_sp_input(batch))
    n = batch["ray_o"].shape[1]
    if n_check is None or n_check >= n:
        n_check = n
        sel = torch.arange(n)
    else:
        sel = torch.linspace(0, n - 1, n_check).long()
    b = {k: v.detach().cpu() for k, v in batch.items()}
    vols_cpu = [v.detach().float().cpu().contiguous() for v in vols]
    sdt = orc.tensor_state_dict(sd)
    rgb_hip = out["rgb_map"][0].cpu()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    errs, sig, tl = [], [], []
    for i in range(0, n_check, chunk):
        idx = sel[i:i + chunk]
        bb = dict(b)
        bb.update(ray_o=b["ray_o"][:, idx], ray_d=b["ray_d"][:, idx], near=b["near"][:, idx], far=b["far"][:, idx])
        with torch.no_grad():
            ref = orc.render(dict(sdt), bb, n_samples=n_samples, training=True, feature_volume=vols_cpu)
        errs.append((rgb_hip[idx] - ref["rgb_map"][0]).abs().max(1).values)
        sig.append(ref["raw"][0].reshape(len(idx), n_samples, 4)[:, -1, 3].clone())
        tl.append(1.0 - ref["weights"][0][:, :-1].sum(1))  # sum of the weights in front of sample i = 1 - T_i
        del ref
    err, sigma_last, t_last = torch.cat(errs), torch.cat(sig), torch.cat(tl)
    ill = sigma_last.abs() < ILL_SIGMA
    raw_hip = out["raw"][0].reshape(n, n_samples, 4)[:, -1, 3]
    worst = int(err.argmax())
    res = {"linf_all": float(err.max()), "linf": float(err[~ill].max()), "n": int(n_check - int(ill.sum())), "n_ill": int(ill.sum()),
           "n_checked": int(n_check), "worst_ray": int(sel[worst]),
           "ill": [{"ray": int(sel[i]), "sigma_last": float(sigma_last[i]), "T_last": float(t_last[i]), "err": float(err[i])}
                   for i in torch.nonzero(ill).reshape(-1)[:list_ill]],
           "ill_linf": float(err[ill].max()) if bool(ill.any()) else 0.0,
           "ill_full_view": int((raw_hip.abs() < ILL_SIGMA).sum()), "rays_full_view": int(n), "ill_sigma": ILL_SIGMA,
           "rays_over_1e-5": int((err > 1e-5).sum()), "rays_over_5e-5": int((err > 5e-5).sum())}
    flipped = ill & (err > 1e-4)
    res["n_flipped"] = int(flipped.sum())  # ill-conditioned rays whose last alpha took the other branch than the oracle's
    fix = getattr(rend, "last_ill", None)
    if fix is not None:  # the march's last-sample fix-up: rays it listed / rays whose last alpha it moved to the other branch
        listed, changed = fix[:2].tolist()
        res["fixup"] = {"listed": int(listed), "changed_side": int(changed), "sigma_band": _ill_consts()[0], "t_min": _ill_consts()[1]}
    # Pass rule: EVERY ray inside the 1e-4 budget.  The only exception is a ray whose last density the oracle itself cannot sign
    # (|sigma_last| < FP32_SIGMA: fp32 rounding of the reference's own formula); it must stay inside its flip bound T_last.
    und = sigma_last.abs() < FP32_SIGMA
    res["n_undecidable"] = int(und.sum())
    res["ok"] = bool((float(err[~und].max()) if bool((~und).any()) else 0.0) <= 1e-4 and bool(((err <= t_last + 1e-4) | ~und).all()))
    return res


def _ill_consts():
    from neuralbody_amd import _lib

    return _lib.ILL_SIGMA, _lib.ILL_T_MIN


def parity_linf(sd, net, rend, batch, n_samples, n_check=4096):
    """(linf over the well-conditioned rays, their count, the count of ill-conditioned rays) — see parity_check."""
    r = parity_check(sd, net, rend, batch, n_samples, n_check)
    return r["linf"], r["n"], r["n_ill"]


def _port_vs_reference():
    """The port timed beside the reference ITSELF on the build container's cores (`--mode cpu-reference`, where /root/reference
    exists): the committed record, or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r04_cpu_baseline_reference.json")) as f:
            r = json.load(f)
        return {"port_over_reference": r["port_over_reference"], "rgb_linf": r["rgb_linf_port_vs_reference_first_chunk"],
                "source": "profiles/r04_cpu_baseline_reference.json (bench.py --mode cpu-reference, %d threads)" % r["cpu_baseline"]["cores"]}
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(sd, bd, vols, n_samples, budget_s=12.0, max_rays=8192):
    """The oracle (CPU restatement of the reference, chunked by 2048 rays like if_clight_renderer.py:107)
    marching a bounded sample of the bench rays through the same feature volumes.  torch's CPU thread
    count is chosen by a short probe (on many-core hosts fewer threads than cores is faster); the count
    used is reported as `cores`."""
    from oracle import neuralbody_oracle as orc

    sdt = orc.tensor_state_dict(sd)
    vols_cpu = [v.detach().cpu().contiguous() for v in vols]  # NCDHW like the reference's .dense()
    n = bd["ray_o"].shape[1]
    sel = torch.linspace(0, n - 1, max_rays).long()
    b = {k: v.detach().cpu() for k, v in bd.items()}

    def run(idx):
        bb = dict(b)
        bb.update(ray_o=b["ray_o"][:, idx], ray_d=b["ray_d"][:, idx], near=b["near"][:, idx], far=b["far"][:, idx])
        t0 = time.perf_counter()
        with torch.no_grad():
            orc.render(dict(sdt), bb, n_samples=n_samples, training=True, feature_volume=vols_cpu)
        return time.perf_counter() - t0

    ncpu = os.cpu_count() or 1
    probe = {}
    for nt in sorted({min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}, reverse=True):  # more than 64 torch threads only thrash
        torch.set_num_threads(nt)
        run(sel[:128])  # warm-up (thread pool, allocator)
        probe[nt] = run(sel[:256])
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    done, t_total = 0, 0.0
    for i in range(0, max_rays, 2048):
        idx = sel[i:i + 2048]
        t_total += run(idx)
        done += len(idx)
        if t_total > budget_s:
            break
    return {"value": done * n_samples / t_total, "unit": "ray-samples/s", "cores": best, "host_cores": ncpu,
            "kind": "port", "rays_per_s": done / t_total, "port_vs_reference": _port_vs_reference(),
            "thread_probe_s_per_256_rays": {str(k): round(v, 3) for k, v in probe.items()},
            "sample": "%d of the bench rays x %d samples, march only (K2-K8) on precomputed feature volumes, "
                      "%.1f s of oracle/neuralbody_oracle.py (torch CPU %s, %d threads)"
                      % (done, n_samples, t_total, torch.__version__, best)}


def cpu_reference_baseline(args, budget_s=15.0, max_rays=8192):
    """`--mode cpu-reference`: the UNMODIFIED reference (lib/networks/renderer/if_clight_renderer.py:62 get_pixel_value in chunks
    of 2048 rays, :107) timed on the host's cores through oracle/ref_harness.py, beside oracle/neuralbody_oracle.py on the same
    rays and the same feature volumes.  Runs WITHOUT a GPU and only where /root/reference exists (the build container): the
    tree cannot travel to the GPU box, where `cpu_baseline.kind` stays "port".  This run pins the port's speed to the
    reference's on the same host: the JSON it prints is committed under profiles/."""
    from tests import synthetic as syn
    from oracle import neuralbody_oracle as orc
    from oracle import ref_harness as rh

    if not rh.available():
        raise SystemExit("--mode cpu-reference needs the reference tree at %s" % rh.REF_ROOT)
    H = W = args.size
    S = args.samples
    sd = syn.make_weights(0, num_train_frame=230)
    body = syn.make_body(seed=0)
    K, R, T = syn.full_coverage_camera(body, H, W)
    ro, rd, near, far, mask = syn.host_image_rays(H, W, K, R, T, body["can_bounds"])
    sel = np.linspace(0, ro.shape[0] - 1, max_rays).astype(np.int64)
    batch = syn.make_batch(body, ro[sel], rd[sel], near[sel], far[sel], mask)
    ns = rh.load()
    ns.cfg.N_samples, ns.cfg.perturb = S, 0
    net = rh.make_reference_network(sd, train_mode=True)
    rend = rh.make_reference_renderer(net)
    bt = rh.torch_batch(batch)
    sdt = orc.tensor_state_dict(sd)
    nt = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(nt)
    with torch.no_grad():
        sp = rend.prepare_sp_input(bt)
        vols = net.encode_sparse_voxels(sp)  # once, untimed: the march is what is measured

        def ref_chunk(i):
            return rend.get_pixel_value(bt["ray_o"][:, i:i + 2048], bt["ray_d"][:, i:i + 2048], bt["near"][:, i:i + 2048],
                                        bt["far"][:, i:i + 2048], vols, sp, bt)["rgb_map"]

        def port_chunk(i):
            bb = dict(bt)
            for k in ("ray_o", "ray_d", "near", "far"):
                bb[k] = bt[k][:, i:i + 2048]
            return orc.render(dict(sdt), bb, n_samples=S, training=True, feature_volume=vols)["rgb_map"]

        out = {}
        for name, fn in (("reference", ref_chunk), ("port", port_chunk)):
            fn(0)[:, :1]  # warm-up
            done, t_total, first = 0, 0.0, None
            for i in range(0, max_rays, 2048):
                t0 = time.perf_counter()
                r = fn(i)
                t_total += time.perf_counter() - t0
                first = r if first is None else first
                done += r.shape[1]
                if t_total > budget_s:
                    break
            out[name] = (done * S / t_total, done, t_total, first)
    diff = float((out["reference"][3].double() - out["port"][3].double()).abs().max())
    v, done, t_total, _ = out["reference"]
    return {"cpu_baseline": {"value": v, "unit": "ray-samples/s", "cores": nt, "host_cores": os.cpu_count(), "kind": "reference",
                             "sample": "%d of the %dx%d bench rays x %d samples, march only on precomputed feature volumes, %.1f s of the "
                                       "reference's Renderer.get_pixel_value (torch CPU %s, %d threads)" % (done, H, W, S, t_total, torch.__version__, nt)},
            "port_value": out["port"][0], "port_over_reference": out["port"][0] / v, "rgb_linf_port_vs_reference_first_chunk": diff}


def cpu_reference_train(args, budget_s=60.0):
    """`--mode cpu-reference-train`: the training step of the UNMODIFIED reference on the host's cores — NetworkWrapper.forward
    (lib/train/trainers/if_nerf_clight.py:18-36: render of 1024 random rays x 64 jittered samples + masked MSE), loss.backward(),
    clip_grad_value_(40), Adam step (lib/train/trainers/trainer.py:46-53) — on the bench scene through oracle/ref_harness.py (the
    dense stand-in for spconv).  The CPU partner of `--mode train` / extras.train_step_ms.  Runs WITHOUT a GPU, only where
    /root/reference exists (the build container); the JSON it prints is committed under profiles/."""
    from tests import synthetic as syn
    from oracle import ref_harness as rh

    if not rh.available():
        raise SystemExit("--mode cpu-reference-train needs the reference tree at %s" % rh.REF_ROOT)
    H = W = args.size
    S = args.samples
    sd = syn.make_weights(0, num_train_frame=230)
    body = syn.make_body(seed=0)
    K, R, T = syn.full_coverage_camera(body, H, W)
    ro, rd, near, far, mask = syn.host_image_rays(H, W, K, R, T, body["can_bounds"])
    g = torch.Generator(device="cpu").manual_seed(0)
    pick = torch.randperm(ro.shape[0], generator=g)[:1024].numpy()
    batch = syn.make_batch(body, ro[pick], rd[pick], near[pick], far[pick], np.ones(1024, bool))
    batch["rgb"] = torch.rand((1, 1024, 3), generator=g).numpy()
    ns = rh.load()
    ns.cfg.N_samples, ns.cfg.perturb, ns.cfg.white_bkgd, ns.cfg.raw_noise_std = S, 1.0, False, 0.0
    net = rh.make_reference_network(sd, train_mode=True)
    wrapper = ns.NetworkWrapper(net)
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    tb = rh.torch_batch(batch)
    nt = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(nt)

    def step():
        ret, loss, stats, _ = wrapper(tb)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_value_(net.parameters(), 40)
        opt.step()
        return float(loss)

    step()  # warm-up (thread pool, allocator, the stand-in's index structures)
    times, loss = [], None
    while sum(times) < budget_s and len(times) < max(args.steps, 1):
        t0 = time.perf_counter()
        loss = step()
        times.append(time.perf_counter() - t0)
    dt = float(np.mean(times))
    return {"metric": "train_step_ms_cpu_reference", "value": dt * 1e3, "unit": "ms", "higher_is_better": False, "steps": len(times),
            "rays_per_step": 1024, "samples_per_ray": S, "ray_samples_per_sec": 1024 * S / dt, "final_loss": loss,
            "train_cpu_baseline": {"value": dt * 1e3, "unit": "ms per step", "cores": nt, "host_cores": os.cpu_count(), "kind": "reference",
                                   "sample": "%d steps of the reference's NetworkWrapper forward + loss.backward() + clip_grad_value_(40) + Adam on "
                                             "1024 random rays x %d samples of the bench scene (torch CPU %s, %d threads; spconv replaced by "
                                             "oracle/spconv_standin.py: dense masked conv3d)" % (len(times), S, torch.__version__, nt)}}


def _train_cpu_record():
    """The committed record of `--mode cpu-reference-train` (the build container's cores), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r05_train_cpu_reference.json")) as f:
            r = json.load(f)["train_cpu_baseline"]
        r["source"] = "profiles/r05_train_cpu_reference.json (bench.py --mode cpu-reference-train in the build container)"
        return r
    except (OSError, KeyError, ValueError):
        return None


def fullview_parity(args, dev):
    """`--mode fullview-parity`: EVERY ray (or --n-check rays) of one timed view against the oracle — the record behind the claim
    that the ill-conditioned-ray diagnostic of parity_check is never needed.  ~2.5 min of CPU for 512 x 512
    x 64."""
    H = W = args.size
    sd, body, net, rend, bd, n_rays = build_scene(dev, H, W, args.samples, args.precision)
    poses = build_poses(dev, body, bd, H, W)
    t0 = time.perf_counter()
    par = parity_check(sd, net, rend, poses[1], args.samples, n_check=args.n_check, list_ill=1 << 20)
    return {"metric": "fullview_parity_rgb_linf", "value": par["linf_all"], "unit": "rgb L-inf vs the CPU oracle", "higher_is_better": False,
            "budget": 1e-4, "precision": net.march_precision(), "seconds": time.perf_counter() - t0,
            "config": {"workload": "synthetic 6890-vertex SMPL scene, %dx%d full-coverage view (pose 1 of the timed cycle), %d samples/ray; "
                                   "%d of %d rays checked" % (H, W, args.samples, par["n_checked"], par["rays_full_view"])},
            "parity": par}


def train_bench(args, dev):
    """Config 4 of BASELINE.json in synthetic form: one training step = NetworkWrapper-style forward (1024 random rays x
    64 jittered samples of the 6890-vertex scene) + MSE loss + backward through decoder and encoder + clip + Adam.
    Informational (the headline metric is the render throughput): prints its own JSON line."""
    from tests import synthetic as syn

    sd, body, net, rend, bd, n_rays = build_scene(dev, args.size, args.size, args.samples, "f32")
    rend.cfg.perturb = 1.0
    g = torch.Generator(device="cpu").manual_seed(0)
    pick = torch.randperm(n_rays, generator=g)[:1024].to(dev)
    tb = dict(bd)
    for k in ("ray_o", "ray_d", "near", "far"):
        tb[k] = bd[k][:, pick].contiguous()
    tb["mask_at_box"] = torch.ones((1, 1024), dtype=torch.bool, device=dev)
    target = torch.rand((1, 1024, 3), generator=g).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)

    def step():
        out = rend.render(tb)
        loss = torch.mean((out["rgb_map"] - target) ** 2)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_value_(net.parameters(), 40)
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    return ({"metric": "train_step_ms", "value": dt * 1e3, "unit": "ms", "higher_is_better": False, "n_gpus": 1,
                      "steps": args.steps, "warmup": args.warmup, "rays_per_step": 1024, "samples_per_ray": args.samples,
                      "ray_samples_per_sec": 1024 * args.samples / dt, "final_loss": float(loss.detach()),
                      # decoder forward + the two backward products = 3 x the forward flops, on the exact-fp32 MFMA kernels; the
                      # step is ~600 launches of 5-30 us and the encoder, so this fraction states how far a 1024-ray step is from
                      # being a matrix-pipe problem at all, not a kernel quality
                      "roofline": {"bound": "mfma", "achieved": 3 * FLOP_PER_SAMPLE * 1024 * args.samples / dt / 1e12,
                                   "peak": PRECISION_INFO["f32"][3], "unit": "TFLOP/s",
                                   "frac": 3 * FLOP_PER_SAMPLE * 1024 * args.samples / dt / 1e12 / PRECISION_INFO["f32"][3],
                                   "traffic": None, "flop_per_step": 3 * FLOP_PER_SAMPLE * 1024 * args.samples,
                                   "note": "whole step (host clock), decoder algebra only; launch- and encoder-bound"},
                      "config": {"workload": "synthetic training step: 1024 random rays, forward + backward (decoder GEMMs and encoder "
                                             "kernels all in libnb_hip.so, no vendor BLAS) + clip_grad_value_(40) + Adam"}})


def turntable_bench(args, dev):
    """Config 2/5 of BASELINE.json in synthetic form: a spiral camera path (gen_path) around the body, every view done
    as nb_raygen -> Renderer.render (encoder + march) -> nb_image_assemble, i.e. finished images on the device; the 4-byte ray
    count of a view is read one view ahead (NovelViewRenderer.render_views).  Informational: prints its own JSON line."""
    from neuralbody_amd import novel_view as nv
    from tests import synthetic as syn

    H = W = args.size
    sd, body, net, rend, bd, n_rays = build_scene(dev, H, W, args.samples, args.precision)
    K, R, T = syn.full_coverage_camera(body, H, W)
    train = []
    for yaw in (0.0, 0.8, 1.6, 2.4):
        _, Rv, Tv = syn.full_coverage_camera(body, H, W, yaw=yaw)
        train.append(np.concatenate([np.concatenate([Rv, Tv.reshape(3, 1)], 1), [[0, 0, 0, 1.0]]], 0))
    path = nv.gen_path(train, args.steps + args.warmup, center=body["world_verts"].mean(0).astype(np.float64))
    frame = {k: v for k, v in bd.items() if k in ("coord", "out_sh", "bounds", "R", "Th", "latent_index")}
    nvr = nv.NovelViewRenderer(rend, H, W, dev, reuse_volumes=args.reuse_volumes)
    from neuralbody_amd import ops

    rays = 0
    for RT in path[:args.warmup]:
        nvr.render_view(K, RT, body["can_bounds"], frame)
    torch.cuda.synchronize()
    ops.MARCH_EVENTS = []
    t0 = time.perf_counter()
    for view in nvr.render_views(((K, RT, body["can_bounds"], frame) for RT in path[args.warmup:]), bgr=True, scale=255.0):
        rays += view["n_rays"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    events, ops.MARCH_EVENTS = ops.MARCH_EVENTS, None
    march_ms = float(np.mean([a.elapsed_time(b) for a, b in events])) if events else float("nan")
    dtype, kernel_name, exec_flop, peak = PRECISION_INFO[net.march_precision()]
    mean_rays = rays / args.steps
    achieved = FLOP_PER_SAMPLE * mean_rays * args.samples / (march_ms * 1e-3) / 1e12
    return ({"metric": "turntable_views_per_sec", "value": args.steps / dt, "unit": "views/s", "higher_is_better": True,
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_view": dt / args.steps * 1e3,
                      "rays_per_sec": rays / dt, "ray_samples_per_sec": rays * args.samples / dt,
                      "mean_rays_per_view": mean_rays, "dtype": dtype, "data": "synthetic",
                      # the march of a spiral view covers fewer rays than the full-coverage bench view (the body's box does not
                      # fill every pixel): algorithmic flops of the rays actually marched / the average march launch (HIP events)
                      "roofline": {"bound": "mfma", "kernel": kernel_name, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                                   "frac": achieved / peak, "traffic": None, "avg_launch_ms": march_ms, "launches": len(events),
                                   "executed_frac": exec_flop * mean_rays * args.samples / (march_ms * 1e-3) / 1e12 / peak},
                      "config": {"workload": "synthetic spiral path, %dx%d, %d samples/ray: raygen + encoder + march + image "
                                             "assembly per view, all on device%s" % (H, W, args.samples, "; frame encoded once" if args.reuse_volumes else "")}})


def extras(args, dev):
    """Informational legs carried by the default JSON line: a short spiral turntable (every view = raygen + encoder + march
    + image assembly, new pose each view) and a short training-step run (config 4 shape)."""
    import copy

    a = copy.copy(args)
    a.steps, a.warmup, a.reuse_volumes = 8, 2, False
    tt = turntable_bench(a, dev)
    # (6 steps behind 2 warm-ups read 6.7-7.5 ms where `--mode train --steps 20 --warmup 5` reads 5.5: the first steps still grow the
    # caching allocator's pools and create Adam's state; with 5 warm-ups the two agree)
    a.steps, a.warmup = 10, 5
    tr = train_bench(a, dev)
    ex = {"turntable_ms_per_view": tt["ms_per_view"], "turntable_rays_per_sec": tt["rays_per_sec"],
          "train_step_ms": tr["value"], "train_ray_samples_per_sec": tr["ray_samples_per_sec"], "train_roofline": tr["roofline"],
          "train_cpu_baseline": _train_cpu_record(),
          "note": "8 spiral views (512x512x64, each view: nb_raygen + encoder + march + nb_image_assemble) / 10 training steps behind 5 warm-ups "
                  "(1024 random rays x 64 jittered samples, forward + backward + clip + Adam); *_ms_per_view / *_march_ms: the timed "
                  "view of this run rendered with the other arithmetics (3 steps each), roofline fraction of each against ITS peak"}
    # the same view in the reference's own precision (exact fp32 MFMA): the record then holds a reference-precision number from
    # the same box
    from neuralbody_amd import ops

    for prec in ("f32",):
        if prec == args.precision:
            continue
        sd, body, net, rend, bd, n_rays = build_scene(dev, args.size, args.size, args.samples, prec)
        with torch.no_grad():
            rend.render(bd)
            torch.cuda.synchronize()
            ops.MARCH_EVENTS = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                rend.render(bd)
            e1.record()
            torch.cuda.synchronize()
            ev, ops.MARCH_EVENTS = ops.MARCH_EVENTS, None
        march = float(np.mean([x.elapsed_time(y) for x, y in ev]))
        peak = PRECISION_INFO[prec][3]
        ex["%s_ms_per_view" % prec] = e0.elapsed_time(e1) / 3
        ex["%s_march_ms" % prec] = march
        ex["%s_roofline_frac" % prec] = FLOP_PER_SAMPLE * n_rays * args.samples / (march * 1e-3) / 1e12 / peak
        # the numerator above is the reference's algebra; the kernel issues fewer flops (merged colour head, latent
        # code as a bias), so the fraction of ITS OWN work over the peak is the occupancy figure — a frac above 1 is accounting
        ex["%s_executed_frac" % prec] = PRECISION_INFO[prec][2] * n_rays * args.samples / (march * 1e-3) / 1e12 / peak
        del net, rend
    ex.update(culled_bench(args, dev))
    ex.update(encoder_bench(args, dev))
    try:
        ex.update(strong_scaling_proxy(args, dev))
    except Exception as e:  # informational leg: never the reason a bench line is lost
        ex["strong8_proxy_error"] = repr(e)
    return ex


def strong_scaling_proxy(args, dev, world=8):
    """What ONE rank of an 8-GPU strong-scaling job does per step, measured on this GPU: the encoder of the next frame on the
    second stream + the march of its 1/8 share of the view (whole 8-row tile bands, parallel.shard_range_tiled) — timed for each
    of the 8 shares in the same fence / render / prefetch loop `bench.py --scaling strong` runs (the all-gather of the RGB tiles is
    only part of it when the bench itself runs under torch.distributed.run: initialising RCCL prints a banner to stdout, which
    would break the one-JSON-line contract of the default run).  `strong8_rank_ms` is the slowest share (the job's step is the MAX over the ranks),
    `strong8_predicted_speedup` = this box's single-GPU step / that: the number the driver's first SCALE record can be read
    against (xGMI wire time for 7 x 0.4 MB per rank is ~20 us and not in it)."""
    import torch.distributed as dist

    from neuralbody_amd import parallel

    H = W = args.size
    sd, body, net, rend, bd, n_rays = build_scene(dev, H, W, args.samples, args.precision)
    poses = build_poses(dev, body, bd, H, W, n_poses=4)
    rend.use_encoder_graph = bool(getattr(args, "encoder_graph", False))
    if True:
        def loop(rng, n_steps, gather):
            tickets = {}
            ev = []
            with torch.no_grad():
                for i in range(n_steps):
                    b = poses[i % len(poses)]
                    fence = rend.fence()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    o = rend.render(b, ray_range=rng, prefetched=tickets.pop(i, None))["rgb_map"][0]
                    tickets[i + 1] = rend.prefetch(poses[(i + 1) % len(poses)], after=fence)
                    if gather:
                        parallel.all_gather_tiles(o)  # (a no-op copy without a process group)
                    e1.record()
                    ev.append((e0, e1))
                torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in ev[2:])
            return ms[len(ms) // 2]

        def loop_once_encoded(rng, n_steps):
            # the turntable case (BASELINE config 5: 144 views of ONE frame): the frame is encoded once, a step is the march of the
            # share alone (NovelViewRenderer.reuse_volumes / Renderer.render(feature_volume=...))
            ev = []
            with torch.no_grad():
                vols = net.encode_sparse_voxels(rend.prepare_sp_input(poses[0]))
                for i in range(n_steps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rend.render(poses[i % len(poses)], ray_range=rng, feature_volume=vols)
                    e1.record()
                    ev.append((e0, e1))
                torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in ev[2:])
            return ms[len(ms) // 2]

        full = loop(None, 8, False)
        shares = [loop(parallel.shard_range_tiled(n_rays, r, world, H, W), 8, dist.is_initialized()) for r in range(world)]
        full_once = loop_once_encoded(None, 8)
        shares_once = [loop_once_encoded(parallel.shard_range_tiled(n_rays, r, world, H, W), 8) for r in range(world)]
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tile = torch.zeros((n_rays // world, 3), device=dev)
        ag = None
        if dist.is_initialized():
            parallel.all_gather_tiles(tile)
            g0.record()
            for _ in range(10):
                parallel.all_gather_tiles(tile)
            g1.record()
            torch.cuda.synchronize()
            ag = g0.elapsed_time(g1) / 10
        out = {"strong8_rank_ms": max(shares), "strong8_share_ms": [round(v, 3) for v in shares], "strong8_full_view_ms": full,
               "strong8_allgather_world1_ms": ag, "strong8_predicted_speedup": full / max(shares),
               "strong8_frame_encoded_once": {"rank_ms": max(shares_once), "full_view_ms": full_once, "predicted_speedup": full_once / max(shares_once),
                                              "note": "the same shares when the frame is encoded ONCE for all views (a turntable of one frame: BASELINE "
                                                      "config 5; Renderer.render(feature_volume=...)): the step is the march of the share alone, the "
                                                      "replicated encoder — the Amdahl term of the figures above — is gone"},
               "strong8_note": "median step of one rank's share (1/8 of the view in whole 8-row tile bands) in the fence / render(ray_range) / "
                               "prefetch loop of --scaling strong, each of the 8 shares in turn on this GPU (all-gather of the RGB tile "
                               "only under torch.distributed.run; DESIGN.md section 6 models it at 0.1-0.15 ms); full_view_ms: the same loop "
                               "over the whole view"}
    return out


def culled_bench(args, dev):
    """The mask-culled renderers the shipped visualisation configs select (if_clight_renderer_mmsk.py / _msk.py: 4 resp. 1
    training-view silhouettes) on the capsule body with synthetic silhouettes, as in the fixtures: ms per view and the share of
    samples that survive the culling."""
    from neuralbody_amd import ops
    from tests import synthetic as syn
    from neuralbody_amd.network import Network
    from neuralbody_amd.renderer import RenderConfig, RendererMmsk, RendererMsk

    H = W = args.size
    sd = syn.make_weights(0, num_train_frame=230)
    body = syn.make_body(seed=0, layout="capsules")
    K, R, T = syn.full_coverage_camera(body, H, W)
    net = Network(num_train_frame=230, precision=args.precision)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    net = net.to(dev).train()
    ro, rd, near, far, mask, n = ops.raygen(H, W, K, R, T, body["can_bounds"], dev)
    n = int(n)
    batch = syn.make_batch(body, np.zeros((1, 3), np.float32), np.zeros((1, 3), np.float32), np.zeros(1, np.float32),
                           np.zeros(1, np.float32), np.ones(1, bool))
    bd = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in batch.items()
          if k not in ("ray_o", "ray_d", "near", "far", "mask_at_box")}
    bd.update(ray_o=ro[None, :n], ray_d=rd[None, :n], near=near[None, :n], far=far[None, :n], mask_at_box=mask[None].bool())
    out = {}
    for kind, nv, cls in (("mmsk", 3, RendererMmsk), ("msk", 1, RendererMsk)):
        msks, Ks, RT = syn.make_view_masks(body, 256, 256, n_views=nv)
        b = dict(bd)
        if kind == "mmsk":
            b.update(msks=torch.from_numpy(msks[None]).to(dev), Ks=torch.from_numpy(Ks[None]).to(dev), RT=torch.from_numpy(RT[None]).to(dev))
        else:
            b.update(msk=torch.from_numpy((msks[0] * 255)[None].astype(np.uint8)).to(dev), K=torch.from_numpy(Ks[0][None]).to(dev),
                     RT=torch.from_numpy(RT[0][None]).to(dev), R0_snap=bd["R"].clone(), Th0_snap=bd["Th"].reshape(1, 3).clone())
        rend = cls(net, RenderConfig(N_samples=args.samples, perturb=0.0, H=H, W=W))
        with torch.no_grad():
            o = rend.render(b, want_raw=True)
            torch.cuda.synchronize()
            ops.MARCH_EVENTS = []
            for _ in range(3):
                rend.render(b)
            torch.cuda.synchronize()
            ev, ops.MARCH_EVENTS = ops.MARCH_EVENTS, None
        out["%s_march_ms" % kind] = float(np.mean([x.elapsed_time(y) for x, y in ev]))
        # a culled sample's raw output is exactly 0 (if_clight_renderer_mmsk.py:54-59); decoded ones are never all-zero
        out["%s_surviving_sample_fraction" % kind] = float((o["raw"][0].abs().sum(-1) != 0).float().mean())
    out["culled_note"] = ("mmsk / msk: RendererMmsk (3 silhouette views) / RendererMsk (1) on the capsule body, %dx%dx%d, 256x256 masks; "
                          "march only; a depth step none of whose 64 samples survives skips its layers" % (H, W, args.samples))
    return out


def encoder_bench(args, dev):
    """Encoder + glue of one view (everything Renderer.render enqueues before the march: prepare_sp_input, the 17 sparse
    layers, the fc_0-folded planes, the latent bias): device time between two HIP events with the march stubbed out, and the
    number of kernel launches counted by torch's profiler."""
    sd, body, net, rend, bd, n_rays = build_scene(dev, args.size, args.size, args.samples, args.precision)

    def front():
        sp = rend.prepare_sp_input(bd)
        vols = net.encode_sparse_voxels(sp)
        net.make_scene(vols, sp, net.march_precision())
        net.latent_bias(sp["latent_index"])

    with torch.no_grad():
        for _ in range(3):
            front()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            front()
        e1.record()
        torch.cuda.synchronize()
        launches = kernel_ms = None
        try:
            from torch.profiler import ProfilerActivity, profile

            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                front()
                torch.cuda.synchronize()
            dev_events = [e for e in prof.events() if e.device_type is not None and str(e.device_type).endswith("CUDA")]
            launches = len(dev_events)
            kernel_ms = sum(e.time_range.end - e.time_range.start for e in dev_events) / 1000.0
        except Exception:  # the profiler is informational
            pass
    return {"encoder_ms": e0.elapsed_time(e1) / 10, "encoder_kernel_ms": kernel_ms, "launches_per_view": launches,
            "encoder_note": "prepare_sp_input + 17 sparse conv/BN/ReLU layers + nb_fold_build + latent bias of one view, no march.  "
                            "encoder_ms: HIP events over 10 back-to-back repetitions with nothing else on the device — set by the launch "
                            "thread (57 launches at ~18 us each), not by the kernels; encoder_kernel_ms: the durations of one repetition's "
                            "kernels and memsets summed (torch.profiler), i.e. what the pass costs behind a march, when the launch thread "
                            "is ahead (profiles/r05_step_timeline.md: the same from a rocprofv3 trace); launches = those events"}


def self_launch(n, argv, dry_run=False):
    """`python bench.py --gpus N` started on its own (no RANK / WORLD_SIZE in the environment): replace this process by
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <argv>` —
    one rank per GPU, the launch the driver's own N > 1 command uses.  Refuses when the box has fewer than N devices."""
    import socket

    if not dry_run:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible on this box (one rank per GPU; RCCL refuses two ranks on "
                             "one device)" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver supports dmabuf IPC only
    sys.stdout.flush()
    sys.stderr.write("[bench] --gpus %d without a launcher: re-executing under torch.distributed.run (port %d)\n" % (n, port))
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def dry_run(args, emit, rank, world):
    """--dry-run: the launch / rendezvous / bookkeeping path of an N-rank run WITHOUT the HIP path (gloo on a box without GPUs:
    the CPU test of the launcher) — process group, barrier-bracketed timed region of stub steps, MAX-over-ranks reduction,
    per-rank rows, one JSON line from rank 0.  It measures nothing and says so (`value` null, `dry_run` true)."""
    import torch.distributed as dist

    from neuralbody_amd.parallel import all_gather_tiles, reduce_timings

    use_gpu = torch.cuda.is_available() and torch.cuda.device_count() >= world
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(dev)
    inited = world > 1 or "RANK" in os.environ
    if inited:
        dist.init_process_group(backend="nccl" if use_gpu else "gloo", init_method="env://",
                                **({"device_id": dev} if use_gpu else {}))
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        time.sleep(0.002 * (rank + 1))  # the stub step: rank r takes 2 (r + 1) ms, so the MAX over ranks is the last rank's
        tiles = all_gather_tiles(torch.full((4, 3), float(rank), device=dev))
        assert tiles.shape[0] == 4 * (world if inited else 1)
    if inited:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed, rows = reduce_timings(elapsed, [2.0 * (rank + 1), 0.0, 2.0 * (rank + 1)], 1, dist.group.WORLD if inited else None, dev)
    if rank == 0:
        emit(json.dumps({"metric": "ray_samples_per_sec", "value": None, "unit": "ray-samples/s", "dry_run": True, "n_gpus": world,
                         "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
                         "scaling": args.scaling, "backend": ("nccl" if use_gpu else "gloo") if inited else None,
                         "per_rank": [{"rank": r, "march_ms": row[0], "allgather_ms": row[1], "median_step_ms": row[2]}
                                      for r, row in enumerate(rows)]}))
    if inited:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)  # >= 3 warm-ups
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--precision", default=None, choices=[None, "auto", "f32", "f16f6"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: one view per GPU per step (views shard with no collective but the tile all-gather); "
                         "strong: one view per step, its rays split over the GPUs")
    ap.add_argument("--prefetch-depth", type=int, default=1, help="how many views ahead the encoder runs on the second stream")
    ap.add_argument("--no-overlap", action="store_true", help="encode and march strictly one after the other on one stream")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational turntable / train-step legs of the JSON line")
    ap.add_argument("--reuse-volumes", action="store_true", help="turntable mode: encode the frame once for all views")
    ap.add_argument("--mode", default="render", choices=["render", "train", "turntable", "cpu-reference", "fullview-parity",
                                                         "cpu-reference-train"])
    ap.add_argument("--n-check", type=int, default=None, help="fullview-parity: rays checked (default: every ray of the view)")
    ap.add_argument("--dry-run", action="store_true", help="launch, rendezvous and cross-rank bookkeeping with stub steps: measures nothing")
    ap.add_argument("--no-strong-leg", action="store_true", help="N > 1 weak runs: skip the strong-scaling leg behind the timed region")
    ap.add_argument("--encoder-graph", action="store_true",
                    help="the prefetched encoder pass as ONE HIP graph launch (Renderer.use_encoder_graph): 1.01 -> 0.67 ms per pass on an idle "
                         "device, 20 us of host time instead of 1 ms; behind a march it changes nothing (profiles/r06_encoder_graph.md)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus, sys.argv[1:], args.dry_run)  # does not return

    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL writes its version banner through C stdio, which a
    # redirected stdout delivers at process exit, behind the JSON line): from here on file descriptor 1 is stderr for everybody, and
    # only `emit` writes to the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def emit(line):
        real_stdout.write(line + "\n")
        real_stdout.flush()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.mode == "cpu-reference":  # the modes without a GPU: they time the reference itself where its tree exists
        emit(json.dumps(cpu_reference_baseline(args)))
        return
    if args.mode == "cpu-reference-train":
        emit(json.dumps(cpu_reference_train(args)))
        return
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (a launcher started %d rank(s): pass the same number as --gpus, or start "
                         "`python bench.py --gpus N` without a launcher — it re-executes itself under torch.distributed.run)"
                         % (args.gpus, world, world))
    if args.dry_run:
        dry_run(args, emit, rank, world)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the HIP path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with one process: exercises RCCL)
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)  # RCCL over xGMI

    from neuralbody_amd import ops
    from neuralbody_amd.parallel import all_gather_tiles

    if args.mode == "train":
        if world != 1:
            raise SystemExit("--mode train is a single-GPU informational run")
        emit(json.dumps(train_bench(args, dev)))
        return
    if args.mode == "fullview-parity":
        emit(json.dumps(fullview_parity(args, dev)))
        return
    if args.mode == "turntable":
        if world != 1:
            raise SystemExit("--mode turntable is a single-GPU informational run (views shard with render_views_sharded)")
        emit(json.dumps(turntable_bench(args, dev)))
        return
    H = W = args.size
    sd, body, net, rend, bd, n_rays = build_scene(dev, H, W, args.samples, args.precision)
    rend.use_encoder_graph = bool(args.encoder_graph)  # the prefetched encoder pass as ONE launch (Renderer._replay_encoder_graph)
    S = args.samples
    poses = build_poses(dev, body, bd, H, W)
    from neuralbody_amd.parallel import render_sharded

    gather_events = []

    overlap = [not args.no_overlap]
    tickets = {}  # view index -> ticket of its encoder pass, args.prefetch_depth views ahead

    def step(i):
        # weak: rank r renders view i + r (N different views per step); strong: every rank works on the SAME view i
        off = rank if args.scaling == "weak" else 0
        b = poses[(i + off) % len(poses)]
        cur = tickets.pop(i, None)
        ahead = i + args.prefetch_depth
        # step i + 1's encoder goes to a second HIP stream (Renderer.prefetch) behind a fence taken before this step's march:
        # every step still encodes one frame and marches one view, the encoder's ~60 small launches run in the march's shadow
        fence = rend.fence() if overlap[0] else None
        if args.scaling == "strong":
            out = render_sharded(rend, b, dist.group.WORLD if dist is not None else None, prefetched=cur)["rgb_map"][0]
            if overlap[0]:
                tickets[ahead] = rend.prefetch(poses[(ahead + off) % len(poses)], after=fence)
            return out
        out = rend.render(b, prefetched=cur)
        if overlap[0]:
            tickets[ahead] = rend.prefetch(poses[(ahead + off) % len(poses)], after=fence)
        if dist is not None:
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
            tiles = all_gather_tiles(out["rgb_map"][0], dist.group.WORLD)
            g1.record()
            gather_events.append((g0, g1))
            return tiles
        return out["rgb_map"][0]

    with torch.no_grad():
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ops.MARCH_EVENTS = []
        step_events = []
        del gather_events[:]
        t0 = time.perf_counter()
        for i in range(args.steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step(args.warmup + i)
            e1.record()
            step_events.append((e0, e1))
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        events, ops.MARCH_EVENTS = ops.MARCH_EVENTS, None
        # N > 1 (any run under a launcher), weak headline: a STRONG leg behind the timed region — the same poses, one view per
        # step with its rays split over the ranks (whole 8-row tile bands) and the RGB tiles all-gathered: north_star's "partition
        # pixel batches across the 8 GPUs".  Its speed-up is quoted against the weak leg's step (one whole view per GPU).
        strong_leg = None
        strong_error = None
        if dist is not None and args.scaling == "weak" and not args.no_strong_leg:
            try:  # the headline above must survive whatever this informational leg does
                stickets = {}

                def strong_step(i):
                    fence = rend.fence() if overlap[0] else None
                    o = render_sharded(rend, poses[i % len(poses)], dist.group.WORLD, prefetched=stickets.pop(i, None))["rgb_map"][0]
                    if overlap[0]:
                        stickets[i + 1] = rend.prefetch(poses[(i + 1) % len(poses)], after=fence)
                    return o

                for i in range(max(args.warmup, 2)):
                    strong_step(i)
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                ops.MARCH_EVENTS = []
                s_events = []
                ts = time.perf_counter()
                for i in range(args.steps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    strong_step(max(args.warmup, 2) + i)
                    e1.record()
                    s_events.append((e0, e1))
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                s_elapsed = time.perf_counter() - ts
                s_march, ops.MARCH_EVENTS = ops.MARCH_EVENTS, None
                share = torch.zeros(((n_rays + world - 1) // world, 3), device=dev)
                all_gather_tiles(share, dist.group.WORLD)
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
                for _ in range(10):
                    all_gather_tiles(share, dist.group.WORLD)
                g1.record()
                torch.cuda.synchronize()
                s_ms = sorted(a.elapsed_time(b) for a, b in s_events)
                strong_leg = {"elapsed": s_elapsed, "local": [float(np.mean([a.elapsed_time(b) for a, b in s_march])) if s_march else float("nan"),
                                                              g0.elapsed_time(g1) / 10, s_ms[len(s_ms) // 2]]}
            except Exception as e:  # noqa: BLE001
                strong_leg, strong_error = None, repr(e)
                ops.MARCH_EVENTS = None
        # the same steps strictly serial on one stream (informational: what a single render() call costs)
        serial_ms = None
        if overlap[0] and dist is None:
            overlap[0] = False
            base = args.warmup + args.steps
            for i in range(args.prefetch_depth + 1):
                step(base + i)  # take the frames still in flight
            base += args.prefetch_depth + 1
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(6):
                step(base + i)
            torch.cuda.synchronize()
            serial_ms = (time.perf_counter() - t1) / 6 * 1e3
            overlap[0] = True
    march_ms = float(np.mean([a.elapsed_time(b) for a, b in events])) if events else float("nan")
    step_ms_in_order = [a.elapsed_time(b) for a, b in step_events]
    step_ms = sorted(step_ms_in_order)
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    per_rank = None
    if dist is not None:
        # MAX over the ranks of the elapsed time; what every rank spent in its march launches and in the RCCL all-gather of the
        # RGB tiles (HIP events on its stream); all ranks must have run the same arithmetic
        from neuralbody_amd import _lib
        from neuralbody_amd.parallel import reduce_timings

        ag_ms = float(np.mean([a.elapsed_time(b) for a, b in gather_events])) if gather_events else float("nan")
        elapsed, rows = reduce_timings(elapsed, [march_ms, ag_ms, step_ms[len(step_ms) // 2]], _lib.PRECISIONS[net.march_precision()],
                                       dist.group.WORLD, dev)
        per_rank = [{"rank": r, "march_ms": row[0], "allgather_ms": row[1], "median_step_ms": row[2]} for r, row in enumerate(rows)]
        # every rank must agree that the leg ran before they meet in its reduction (a rank that raised would leave the others waiting)
        if not args.no_strong_leg and args.scaling == "weak":
            okf = torch.tensor([1.0 if strong_leg is not None else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN, group=dist.group.WORLD)
            if float(okf.item()) < 1.0:
                strong_leg = {"error": strong_error or "another rank's strong leg failed"}
        if strong_leg is not None and "error" not in strong_leg:
            s_el, s_rows = reduce_timings(strong_leg["elapsed"], strong_leg["local"], _lib.PRECISIONS[net.march_precision()], dist.group.WORLD, dev)
            strong_leg = {"ms_per_step": s_el / args.steps * 1e3, "steps": args.steps,
                          "value": n_rays * S * args.steps / s_el, "unit": "ray-samples/s (one view per step, rays split over the ranks)",
                          "measured_speedup_vs_one_gpu_view": (elapsed / args.steps) / (s_el / args.steps),
                          "per_rank": [{"rank": r, "march_ms": row[0], "allgather_ms": row[1], "median_step_ms": row[2]}
                                       for r, row in enumerate(s_rows)],
                          "note": "strong leg behind the timed (weak) region: the same poses, one view per step, every rank encodes the frame "
                                  "and marches its whole 8-row tile bands, one RCCL all-gather of the RGB tiles per view (allgather_ms: that "
                                  "collective alone, 10 back to back); measured_speedup = the weak leg's ms_per_step (one whole view per GPU, "
                                  "all GPUs busy) / this leg's ms_per_step — the figure north_star's >= 6x at 8 GPUs refers to; a 1-GPU run's "
                                  "extras.strong8_predicted_speedup is its single-GPU proxy"}

    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot be
    # read from inside the process); the committed summary is quoted when it matches the workload
    traffic = None
    tfile = {"f16f6": "r06_march_fold_traffic.json"}.get(net.march_precision())
    tpath = os.path.join(ROOT, "profiles", tfile or "none")
    if tfile and (H, W, S) == (512, 512, 64) and args.scaling == "weak" and os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f)["hbm_bytes_per_launch"]
    views_per_step = world if args.scaling == "weak" else 1
    rays_per_launch = n_rays if args.scaling == "weak" else (n_rays + world - 1) // world
    total_rays = n_rays * views_per_step * args.steps
    samples_per_s = total_rays * S / elapsed
    dtype, kernel_name, exec_flop, peak = PRECISION_INFO[net.march_precision()]
    achieved_tflops = FLOP_PER_SAMPLE * rays_per_launch * S / (march_ms * 1e-3) / 1e12
    result = {
        "metric": "ray_samples_per_sec", "value": samples_per_s, "unit": "ray-samples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "median_ms_per_step": median_ms, "step_ms": [round(v, 3) for v in step_ms_in_order[:32]], "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": dtype,
        "data": "synthetic",
        "rays_per_sec": total_rays / elapsed,
        "config": {"workload": "synthetic 6890-vertex SMPL scene, %dx%d full-coverage view, %d samples/ray, "
                               "Renderer.render = encoder + fused march, %s; the timed region cycles through %d camera poses; %s" % (
                                   H, W, S, "one view per GPU per step" if args.scaling == "weak" else "one view per step, rays split over the GPUs", len(poses),
                                   "every step encodes one frame and marches one view, the encoder of step i + 1 enqueued on a second HIP stream "
                                   "before the march of step i (Renderer.prefetch; serial_ms_per_step: the same steps on one stream)"
                                   if overlap[0] else "encoder and march one after the other on one stream"),
                   "encoder_overlap": bool(overlap[0]), "encoder_graph": bool(getattr(rend, "use_encoder_graph", False)),
                   "rays_per_view": n_rays, "out_sh": [int(s) for s in body["out_sh"]],
                   "arithmetic": {"f32": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32), trilinear gather on the VALU",
                                  "f16f6": "fc_0 folded into the volume (U = fc_0 . V per active voxel, fp16 head + remainder; the "
                                           "trilinear lookup is an MFMA against the sparse weight matrix of the workgroup's voxel "
                                           "list, three fp16 products, fp32 accumulate); fc_1, fc_2 and the colour head: fp16 head x "
                                           "fp16 head on v_mfma_f32_32x32x16_f16 + the two head x remainder cross terms in 4 x 6 bits "
                                           "(fp4 e2m1 weights, bf6 e3m2 activations, E8M0 scales per 32 K) on "
                                           "v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulate; four waves share 64 rays, activations in "
                                           "LDS, weights streamed from L2, two workgroups per CU"}[net.march_precision()],
                   "parallelism": "views/rays sharded across %d GPU(s)%s" % (world, ", RCCL all-gather of RGB tiles" if world > 1 else "")},
        "roofline": {"bound": "mfma", "kernel": kernel_name, "achieved": achieved_tflops,
                     "peak": peak, "unit": "TFLOP/s", "frac": achieved_tflops / peak,
                     "traffic": traffic, "traffic_unit": "bytes/launch, QUOTED from the committed PMC record of this kernel and workload (%s: separate rocprofv3 --pmc passes, "
                                                        "tools/pmc_traffic.sh), not measured by this run" % os.path.relpath(tpath, ROOT),
                     "avg_launch_ms": march_ms,
                     "executed_tflops": exec_flop * rays_per_launch * S / (march_ms * 1e-3) / 1e12,
                     "executed_frac": exec_flop * rays_per_launch * S / (march_ms * 1e-3) / 1e12 / peak,
                     "note": "achieved = 859904 algorithmic flop/sample x %d samples/launch / avg launch time (HIP events); "
                             "the kernel issues ~%.0f MFMA flop/sample (fp16-equivalent pipe time; fc_0 folded into the volume, merged "
                             "colour head), so executed_frac is the matrix-pipe occupancy; compulsory HBM traffic is ~150 MB/launch "
                             "(<0.1%% of the launch time at 8 TB/s)" % (rays_per_launch * S, exec_flop)},
    }
    if serial_ms is not None:
        result["serial_ms_per_step"] = serial_ms
    if per_rank is not None:
        result["per_rank"] = per_rank
    if isinstance(strong_leg, dict) and ("ms_per_step" in strong_leg or "error" in strong_leg):
        result["strong_leg"] = strong_leg
    if net.precision == "auto":
        result["config"]["auto"] = {"chosen": net.march_precision(), "six_bit_small_fraction_worst_layer": net._auto[2] if net._auto else None}
    if rank == 0 and world == 1 and not args.no_extras:
        result["extras"] = extras(args, dev)  # before the CPU legs: their thread pools compete with the launch thread
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        par = parity_check(sd, net, rend, poses[1], S)
        result["parity_linf"] = par["linf"]
        result["parity_linf_all"] = par["linf_all"]
        result["parity"] = par
        result["parity_note"] = ("rgb L-inf of %d rays of a timed view vs the CPU oracle (same feature volumes), budget 1e-4: parity_linf_all "
                                 "is over ALL of them and is what parity.ok holds to the budget.  The reference's 1e10 last interval makes a "
                                 "ray's last alpha a step function of the sign of its last density; the march lists the rays whose last density "
                                 "it cannot sign (parity.fixup: listed / changed_side over the whole view) and recomputes those at fp32 level "
                                 "(nb_march ill_scratch).  parity.ill lists the %d checked ray(s) within %g of the step in the oracle with "
                                 "their density, T_last and measured error; parity.n_undecidable counts rays within the oracle's own fp32 "
                                 "rounding (%g) of it, which are bounded by T_last instead"
                                 % (par["n"] + par["n_ill"], par["n_ill"], ILL_SIGMA, FP32_SIGMA))
        with torch.no_grad():
            vols = net.encode_sparse_voxels(rend.prepare_sp_input(poses[0]))
        result["cpu_baseline"] = cpu_baseline(sd, poses[0], vols, S)
    if rank == 0:
        emit(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
