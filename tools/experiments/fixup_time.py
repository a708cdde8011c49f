"""Duration of nb_march_fixup_kernel behind a bench march (torch.profiler device events)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from torch.profiler import ProfilerActivity, profile
dev = torch.device("cuda:0")
sd, body, net, rend, bd, n = bench.build_scene(dev, 512, 512, 64, None)
with torch.no_grad():
    for _ in range(3):
        rend.render(bd)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            rend.render(bd)
        torch.cuda.synchronize()
ev = [e for e in prof.events() if "fixup" in e.name]
print("nb_march_fixup_kernel: %d launches, %.1f us average; listed %d" % (len(ev), sum(e.time_range.end - e.time_range.start for e in ev) / max(len(ev), 1), int(rend.last_ill[0])))
