import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers as H
from tests.golden import scenes
DEV = "cuda:0"
r, sd, body, batch, cam, _ = scenes.build("small")
for scale in (1e2, 1e4, 3e5):
    big = dict(sd)
    big["fc_0.weight"] = (np.array(sd["fc_0.weight"]) * scale).astype(np.float32)
    big["fc_0.bias"] = (np.array(sd["fc_0.bias"]) * scale).astype(np.float32)
    bd = H.device_batch(batch, DEV)
    outs = {}
    for prec in ("f32", "f16f6"):
        net = H.make_network(big, DEV, True, prec)
        with torch.no_grad():
            outs[prec] = H.make_renderer(net, r).render(bd, want_raw=True) if False else H.make_renderer(net, r).render(bd)
        o = outs[prec]
        print("scale %g %-7s" % (scale, prec), {k: int((~torch.isfinite(v)).sum()) for k, v in o.items()},
              "rgb diff vs f32 %.3g" % float((o["rgb_map"] - outs["f32"]["rgb_map"]).abs().nan_to_num(9.0).max()))
