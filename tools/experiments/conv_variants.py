"""One encoder convolution launch (the first nb_enc_conv16 call of the given channel pair in a view's encoder pass) timed under several
library builds in ONE process: the arguments of the product's own call are captured, then every variant library is opened beside
the product's and called with them.
    python tools/experiments/conv_variants.py 64:64,128:128 _s1a1 _s1a2 ...        (suffixes of neuralbody_amd/lib/libnb_hip<suffix>.so)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from neuralbody_amd import _lib, ops  # noqa: E402

pairs = [tuple(int(x) for x in p.split(":")) for p in sys.argv[1].split(",")]
suffixes = [""] + sys.argv[2:]
dev = torch.device("cuda:0")
sd, body, net, rend, bd, n = bench.build_scene(dev, 512, 512, 64, "f16f6")
captured = {}
real_conv16 = ops.enc_conv16


def spy(in_split, in_grid, in_dhw, out_lin, n_out, n_out_max, out_dhw, stride, wpacked, cin, cout, stats=None, bf16=False):
    out = real_conv16(in_split, in_grid, in_dhw, out_lin, n_out, n_out_max, out_dhw, stride, wpacked, cin, cout, stats=stats, bf16=bf16)
    if (cin, cout) in pairs and (cin, cout) not in captured:  # the tensors stay referenced: their memory is not handed out again
        captured[(cin, cout)] = dict(t=(in_split, in_grid, out_lin, n_out, wpacked, out[0], out[1].clone()), in_dhw=in_dhw, out_dhw=out_dhw,
                                    n_out_max=n_out_max, stride=stride, bf16=bf16)
    return out


ops.enc_conv16 = spy
with torch.no_grad():
    sp = rend.prepare_sp_input(bd)
    vols = net.encode_sparse_voxels(sp)
torch.cuda.synchronize()
ops.enc_conv16 = real_conv16
real = _lib.lib().nb_enc_conv16
for key in pairs:
    if key not in captured:
        print("no call for", key)
        continue
    c = captured[key]
    in_split, in_grid, out_lin, n_out, wpacked, out_rows, stats = c["t"]
    a = (ops.ptr(in_split), int(in_split.shape[1]), ops.ptr(in_grid), ops._i3(c["in_dhw"]), ops.ptr(out_lin), ops.ptr(n_out), int(c["n_out_max"]),
         ops._i3(c["out_dhw"]), int(c["stride"]), ops.ptr(wpacked), key[0], key[1], ops.ptr(out_rows), ops.ptr(stats), 2 if c["bf16"] else 0,
         ops._stream())
    print("== %d -> %d, %d rows (capacity %d)" % (key[0], key[1], int(n_out.item()), c["n_out_max"]))
    for sfx in suffixes:
        path = os.path.join(ROOT, "neuralbody_amd", "lib", "libnb_hip%s.so" % sfx)
        h = C.CDLL(path)
        f = h.nb_enc_conv16
        f.restype, f.argtypes = real.restype, real.argtypes
        for _ in range(3):
            f(*a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f(*a)
        e1.record()
        torch.cuda.synchronize()
        print("  %-10s %7.1f us" % (sfx or "(product)", e0.elapsed_time(e1) * 1000 / 20))
        if "_t" in sfx:  # stamped build (-DNB_CONV_TIMING=1): wave 0 of every 16th workgroup, cycle counter at the phase boundaries
            import numpy as np
            cap_rows = (int(c["n_out_max"]) + 127) // 128 * 128
            nwg = (int(n_out.item()) + 127) // 128
            ks = list(range(0, (nwg + 15) // 16))
            t = np.stack([out_rows[cap_rows - 160 - k2].view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff for k2 in ks])[:, :62]
            d = lambda x, y: float(((t[:, y] - t[:, x]) & 0xffffffff).mean())
            issue = np.mean([d(2 if o == 0 else 2 + 2 * o, 3 + 2 * o) for o in range(27)])
            wait = np.mean([d(3 + 2 * o, 4 + 2 * o) for o in range(26)])
            print("      stamps of %d workgroups: entry -> indices %.0f, -> first operands %.0f, per offset: multiply %.0f + wait/barrier %.0f, "
                  "loop %.0f, stores + sums %.0f, atomics issued %.0f, all landed %.0f; total %.0f cycles" % (
                      len(ks), d(0, 1), d(1, 2), issue, wait, d(2, 56), d(56, 59), d(59, 60), d(60, 61), d(0, 61)))
            print("      start of the workgroups relative to the first: " + " ".join("%d" % ((x - t[0, 0]) & 0xffffffff) for x in t[:, 0]))
