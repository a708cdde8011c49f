#!/bin/bash
# times the march kernel (HIP events inside bench.py) for every experiment build lib/libnb_hip_<tag>.so
for so in neuralbody_amd/lib/libnb_hip*.so; do
  tag=$(basename $so .so)
  NB_PRECISION=${NB_PRECISION:-bf16x3} NB_LIB_PATH=$so python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$tag', 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f ms' % j['ms_per_step'])
"
done
