"""Encoder + fold planes of one view, alone on the device (bench.encoder_bench) and the training step: A/B of library variants.
NOTE: the back-to-back time is set by the launch thread (57 launches x ~18 us), it does not see kernel changes; compare the kernels' own
durations (encoder_kernel_ms, or a rocprofv3 timeline).
    NB_LIB_PATH=neuralbody_amd/lib/libnb_hip_<variant>.so python tools/experiments/encoder_time.py [train]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

a = argparse.Namespace(size=512, samples=64, precision=os.environ.get("NB_BENCH_PRECISION"), steps=20, warmup=5)
dev = torch.device("cuda:0")
e = bench.encoder_bench(a, dev)
print("encoder %.4f ms back to back (launch-bound), %s ms of kernels, %s launches" % (e["encoder_ms"], e.get("encoder_kernel_ms"), e["launches_per_view"]))
if "train" in sys.argv[1:]:
    t = bench.train_bench(a, dev)
    print("train step %.3f ms" % t["value"])
