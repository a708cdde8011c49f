"""Per-layer check of the M-split f16f6 march kernel (nb_march_ms6.hip) against nb_decode_points' exact-fp32 activation tap.

Needs a debug build of the library:
    NB_EXTRA_FLAGS=-DMS6_TAP NB_LIB_SUFFIX=_tap python -m neuralbody_amd.build
    NB_LIB_PATH=neuralbody_amd/lib/libnb_hip_tap.so python tools/experiments/ms6_tap_check.py
In that build workgroup 0 writes, at depth step 0, the accumulators of fc_0, fc_1, fc_2 (pre-activation) and of the folded
view layer as [layer][feature][sample] into the `raw` output.  The same 64 sample points go through nb_decode_points with the
debug tap (F | h1 | h2 | h3 | G | V | PE, post-relu) and the two are compared layer by layer.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from neuralbody_amd import ops  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.golden import scenes  # noqa: E402


def main():
    dev = "cuda:0"
    r, sd, body, batch, cam, t_rand = scenes.build("small")
    net = H.make_network(sd, dev, True, precision="f16f6")
    bd = H.device_batch(batch, dev)
    rend = H.make_renderer(net, r)
    with torch.no_grad():
        sp = rend.prepare_sp_input(bd)
        vols = net.encode_sparse_voxels(sp)
        scene = net.make_scene(vols, sp)
        lb = net.latent_bias(sp["latent_index"])
        ray_o, ray_d = bd["ray_o"][0].contiguous(), bd["ray_d"][0].contiguous()
        near, far = bd["near"][0].contiguous(), bd["far"][0].contiguous()
        S = r["n_samples"]
        t_vals = torch.linspace(0.0, 1.0, steps=S).to(dev)
        out = ops.march(scene, net.packed_weights("f16f6"), lb, ray_o, ray_d, near, far, t_vals, want_raw=True, precision="f16f6")
        torch.cuda.synchronize()
        tap = out["raw"].reshape(-1)[: (3 * 256 + 128) * 64].cpu().numpy()
        layers = [tap[i * 256 * 64:(i + 1) * 256 * 64].reshape(256, 64) for i in range(3)]
        view = tap[3 * 256 * 64:].reshape(128, 64)
        # the same 64 points through the exact-fp32 point decoder
        z0 = near[:64] * (1.0 - t_vals[0]) + far[:64] * t_vals[0]
        pts = (ray_o[:64] + ray_d[:64] * z0[:, None]).contiguous()
        vd = (ray_d[:64] / ray_d[:64].norm(dim=-1, keepdim=True)).contiguous()
        raw32, dbg = ops.decode_points(scene, net.packed_weights("f32"), lb, pts, vd, debug=True, precision="f32")
        dbg = dbg.cpu().numpy()
    names = ["fc_0 (h1)", "fc_1 (h2)", "fc_2 (h3)"]
    offs = [352, 608, 864]
    ok = True
    for i in range(3):
        ref = dbg[:, offs[i]:offs[i] + 256].T  # [feature, sample], post relu
        got = np.maximum(layers[i], 0.0)
        err = np.abs(got - ref)
        print("%-10s max |err| %.3e (ref max %.3e)  worst feature %d sample %d" % (
            names[i], err.max(), np.abs(ref).max(), *np.unravel_index(err.argmax(), err.shape)))
        if err.max() > 1e-3 * max(1.0, np.abs(ref).max()):
            ok = False
            bad_f = np.where(err.max(1) > 1e-3)[0]
            bad_s = np.where(err.max(0) > 1e-3)[0]
            print("   bad features (%d): %s" % (len(bad_f), bad_f[:40]))
            print("   bad samples  (%d): %s" % (len(bad_s), bad_s[:40]))
    ref = dbg[:, 1376:1376 + 128].T
    got = np.maximum(view, 0.0)
    err = np.abs(got - ref)
    print("%-10s max |err| %.3e (ref max %.3e)" % ("view (V)", err.max(), np.abs(ref).max()))
    if err.max() > 1e-3 * max(1.0, np.abs(ref).max()):
        ok = False
        print("   bad features: %s" % np.where(err.max(1) > 1e-3)[0][:40])
        print("   bad samples : %s" % np.where(err.max(0) > 1e-3)[0][:40])
    # final raw of step 0 is not available in the tap build; compare rgb of the full render instead
    ref_full = H.golden("small")
    print("rgb_map L-inf vs reference fixture: %.3e" % np.abs(out["rgb_map"].cpu().numpy() - ref_full["rgb_map"][0]).max())
    print("TAP CHECK", "OK" if ok else "FAILED")


if __name__ == "__main__":
    main()
