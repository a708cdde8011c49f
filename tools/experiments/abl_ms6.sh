#!/bin/bash
# ablation / variant builds of the M-split f16f6 march kernel (results of the ABL builds are wrong, only the timing is meaningful):
#   tools/experiments/abl_ms6.sh build "<flags1>" "<flags2>" ...   (here; tag = flags with -D / MS6_ stripped)
#   tools/experiments/abl_ms6.sh run                                (on the GPU box)
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  shift
  for v in "$@"; do
    tag=$(echo "$v" | sed 's/-D//g; s/MS6_//g; s/NB_//g; s/ABL_//g; s/[ =]/_/g'); tag=${tag:-base}
    NB_EXTRA_FLAGS="$v" NB_LIB_SUFFIX=_ms6$tag python -m neuralbody_amd.build > /dev/null 2>&1 &
  done
  wait; ls neuralbody_amd/lib/libnb_hip_ms6*.so
else
  for so in neuralbody_amd/lib/libnb_hip_ms6*.so; do
    tag=$(basename $so .so)
    NB_LIB_PATH=$PWD/$so timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$tag', 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'], 'parity', j.get('parity_linf'))
"
  done
fi
