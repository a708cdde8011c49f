"""How much of the bench view is provably dead?  A sample whose transmittance T is EXACTLY 0 in fp32 contributes
exactly nothing (weights = alpha * T), so skipping it is bit-identical to the reference."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
for name in ("box", "capsules"):
    sd, body, net, rend, bd, n = bench.build_scene(dev, 512, 512, 64, None)
    out = rend.render(bd)
    w = out["weights"][0]                      # [n, S]
    acc = out["acc_map"][0]
    # transmittance before sample k: T_k = prod_{j<k} (1 - alpha_j + 1e-10); alpha_j = w_j / T_j
    order = rend._tile_order(bd, n, 0, n).long()
    S = w.shape[1]
    raw = rend.render(bd, want_raw=True) if False else None
    # recompute T from the weights: w_k = alpha_k T_k  and T_{k+1} = T_k (1 - alpha_k + 1e-10)
    T = torch.ones(n, device=dev)
    dead = torch.zeros((n, S), dtype=torch.bool, device=dev)
    for k in range(S):
        dead[:, k] = T == 0
        alpha = torch.where(T > 0, w[:, k] / T, torch.ones_like(T))
        T = T * (1.0 - alpha + 1e-10)
    print("rays %d: dead (T == 0 exactly) samples %.1f%%; acc mean %.3f" % (n, 100 * dead.float().mean().item(), acc.mean().item()))
    for gsz in (64, 128):
        d = dead[order][: (n // gsz) * gsz].view(-1, gsz, S).all(1)
        print("  groups of %d rays: %.1f%% of the (group, step) pairs are all-dead" % (gsz, 100 * d.float().mean().item()))
    tiny = torch.zeros((n, S), dtype=torch.bool, device=dev)
    T = torch.ones(n, device=dev)
    for k in range(S):
        tiny[:, k] = T < 1e-6
        alpha = torch.where(T > 0, w[:, k] / T, torch.ones_like(T))
        T = T * (1.0 - alpha + 1e-10)
    d = tiny[order][: (n // 128) * 128].view(-1, 128, S).all(1)
    print("  T < 1e-6 for all 128 rays: %.1f%% of the pairs" % (100 * d.float().mean().item()))
    break
