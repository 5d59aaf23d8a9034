#!/bin/bash
# round 6: cjxl-shaped jobs of 256 through the wave-wide LF kernel (JXL_HIP_WIDE_WP lifts the round-4 exclusion), one or two LF groups per wavefront (JXL_HIP_LF_FORCE_BIG -1 / 0)
# CFGS: "wide_first in_flight chain(0/1) force_big" ;-separated
export JXL_HIP_WIDE_WP=1
IFS=';' read -ra L <<< "${CFGS:-4 11 1 -1;4 11 0 -1;4 11 1 0;100000 4 0 -1;100000 6 0 -1;100000 11 0 -1}"
for cfg in "${L[@]}"; do
  set -- $cfg
  if [ "$3" = "0" ]; then export JXL_HIP_NO_WIDE_CHAIN=1; else unset JXL_HIP_NO_WIDE_CHAIN; fi
  export JXL_HIP_LF_FORCE_BIG=$4
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-realistic --no-extras --no-cpu-baseline --main-tree-shape 1 --main-texture 5 --wide-first $1 --in-flight $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'wide_first': $1, 'in_flight': $2, 'chain': $3, 'force_big': $4, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'steady': d.get('steady_state_ms_per_step'), 'stage_ms': d.get('stage_ms'), 'first_steps_ms': d['step_end_ms'][:4]}))"
done
