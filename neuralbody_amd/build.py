"""Builds libnb_hip.so (gfx950) in-tree with hipcc.  No CMake, no JIT cache: the .so sits next
to the package so it travels with a snapshot of the repository.

    python -m neuralbody_amd.build [--force]
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
# experiments: NB_EXTRA_FLAGS="-DNB_ABL_..." NB_LIB_SUFFIX=_abl builds lib/libnb_hip_abl.so (see _lib.py NB_LIB_PATH)
SUFFIX = os.environ.get("NB_LIB_SUFFIX", "")
LIB_PATH = os.path.join(LIB_DIR, "libnb_hip%s.so" % SUFFIX)
STAMP = os.path.join(LIB_DIR, "libnb_hip%s.stamp" % SUFFIX)
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wno-unused-result"] + os.environ.get("NB_EXTRA_FLAGS", "").split()


# per-file flags.  nb_march_fold.hip: hipcc's SLP vectoriser packs the gather's and the heads' scalar fp32 FMAs into v_pk_fma_f32,
# which on gfx950 costs an order of magnitude more issue time than the two v_fma_f32 it replaces (MI355X_MICROARCH.md,
# "price of one filler": +22 cycles per v_pk_fma_f32) and needs aligned register pairs (33 spilled registers with, 0 without)
FILE_FLAGS = {"nb_march_fold.hip": ["-fno-slp-vectorize", "-Wno-inline-asm"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    for p in sources() + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + \
            [os.path.join(ROOT, "include", "nb_hip.h")]:
        h.update(p.encode())
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()


def is_fresh():
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as f:
        return f.read().strip() == _digest()


def build(force=False, verbose=True):
    """Compile every .hip under csrc/ and link libnb_hip.so.  hipcc cross-compiles for gfx950
    without a GPU present.  Objects are compiled in parallel (one hipcc process per file)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    if not force and is_fresh():
        if verbose:
            print("[nb build] up to date:", LIB_PATH)
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src)[:-4] + SUFFIX + ".o")
        objs.append(obj)
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print("[nb build]", " ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB_PATH] + objs  # no vendor BLAS: every GEMM is ours
    if verbose:
        print("[nb build]", " ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(_digest())
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
