"""Register / scratch / LDS figures of the kernels of one HIP source, from the code object's metadata notes:

    python tools/kernel_meta.py neuralbody_amd/csrc/nb_march_fold.hip [substring of the kernel name]

Compiles the file for gfx950 with the product's flags (neuralbody_amd/build.py: FLAGS + FILE_FLAGS, device side only) and prints,
per kernel: vgpr / agpr / sgpr counts, spilled registers, private (scratch) segment, group (LDS) segment."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuralbody_amd import build as nb  # noqa: E402

KEYS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
        ".group_segment_fixed_size")


def main():
    src = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, "k.co")
        cmd = ["/opt/rocm/bin/hipcc"] + nb.FLAGS + nb.FILE_FLAGS.get(os.path.basename(src), []) + ["--cuda-device-only", "--no-gpu-bundle-output", "-c", src, "-o", co]
        subprocess.check_call(cmd)
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout

    def demangle(n):
        try:
            return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
        except OSError:
            return n

    cur = {}
    rows = []
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*(\.[a-z_]+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == ".agpr_count" and cur.get(".name"):  # first key of the next kernel's map in llvm's ordering
            rows.append(cur)
            cur = {}
        if k in KEYS or k == ".name":
            cur[k] = v
    if cur:
        rows.append(cur)
    seen = set()
    for r in rows:
        name = demangle(r.get(".name", "?"))
        if pat not in name or name in seen or ".vgpr_count" not in r:
            continue
        seen.add(name)
        print("%s\n    vgpr %s  agpr %s  sgpr %s  vgpr_spill %s  sgpr_spill %s  private_segment %s B  lds %s B" % (
            name, r.get(".vgpr_count"), r.get(".agpr_count"), r.get(".sgpr_count"), r.get(".vgpr_spill_count"),
            r.get(".sgpr_spill_count"), r.get(".private_segment_fixed_size"), r.get(".group_segment_fixed_size")))


if __name__ == "__main__":
    main()
