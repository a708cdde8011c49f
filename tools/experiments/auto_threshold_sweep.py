"""precision 'auto' falls back to the exact kernel when a layer's share of "small" weights (below 1/8 of their (row, 32 K) block maximum)
exceeds SIX_BIT_MAX_SMALL.  That threshold was set for fp6 e2m3 cross weights; with fp4 e2m1 (round 6) this sweep re-measures what it
protects: per-column gains of 2^-g .. 2^g on fc_1 (compensated on fc_0: the function is unchanged) widen the blocks' dynamic range;
for every g: the statistic and the f16f6 render's rgb error against the f32 render of the same weights (small fixture, 807 rays)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuralbody_amd import ops
from tests import helpers as H, synthetic as syn
from tests.golden import scenes

DEV = "cuda:0"
r, sd0, body, batch, cam, _ = scenes.build("small")
bd = H.device_batch(batch, DEV)
print("| gain range 2^+-g | small fraction (fc_1, fc_2, colour head) | rgb L-inf f16f6 vs f32 | 'auto' picks |")
print("|---|---|---|---|")
for g in (0, 1, 2, 3, 4, 5, 6):
    rs = np.random.RandomState(5)
    gain = np.exp2(rs.uniform(-g, g, 256)).astype(np.float32) if g else np.ones(256, np.float32)
    sd = dict(sd0)
    sd["fc_0.weight"] = (np.array(sd0["fc_0.weight"]) * gain[:, None, None]).astype(np.float32)
    sd["fc_0.bias"] = (np.array(sd0["fc_0.bias"]) * gain).astype(np.float32)
    sd["fc_1.weight"] = (np.array(sd0["fc_1.weight"]) / gain[None, :, None]).astype(np.float32)
    outs = {}
    for prec in ("f32", "f16f6"):
        net = H.make_network(sd, DEV, True, prec)
        rend = H.make_renderer(net, r)
        with torch.no_grad():
            outs[prec] = rend.render(bd)["rgb_map"]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        neta = H.make_network(sd, DEV, True, "auto")
        pick = neta.march_precision()
    frac = ops.six_bit_small_fraction(neta.packed_weights("f16f6")).cpu().numpy()
    print("| %d | %s | %.2e | %s |" % (g, np.round(frac, 3), float((outs["f16f6"] - outs["f32"]).abs().max()), pick))
