#!/bin/bash
# build the ablation variants of the f16f8 march kernel (results are wrong, only the timing is meaningful):
#   tools/experiments/abl_f16.sh build      (here)   then on the GPU box:   tools/experiments/abl_f16.sh run
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  for v in ${ABL_SET:-"" NOGATHER NOCONV NOPE NOX NOM "NOX -DF_ABL_NOM" NOLDS NODMA NOBAR "NOGATHER -DF_ABL_NOCONV -DF_ABL_NOPE"}; do
    tag=$(echo "$v" | sed 's/ -DF_ABL_/_/g'); tag=${tag:-base}
    flags=""; [ -n "$v" ] && flags="-DF_ABL_$v"
    NB_EXTRA_FLAGS="$flags" NB_LIB_SUFFIX=_f16$tag python -m neuralbody_amd.build > /dev/null 2>&1 &
  done
  wait; ls neuralbody_amd/lib/libnb_hip_f16*.so
else
  for so in neuralbody_amd/lib/libnb_hip_f16*.so; do
    tag=$(basename $so .so)
    NB_PRECISION=f16f8 NB_LIB_PATH=$so python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$tag', 'march %.2f ms' % j['roofline']['avg_launch_ms'])
"
  done
fi
