#!/bin/bash
# round 6: the cjxl-shaped workload (weighted-predictor LF trees: SIMT LF launch 510-577 ms per 256 frames, the wave-wide kernel 416 ms in the capped four-groups-per-workgroup launch)
# with the cold-start LF stages chained / free / absent.  CFGS: "wide_first in_flight chain(0/1)" triples
IFS=';' read -ra L <<< "${CFGS:-4 11 1;4 11 0;0 11 0;2 11 0;1 11 0}"
for cfg in "${L[@]}"; do
  set -- $cfg
  if [ "$3" = "0" ]; then export JXL_HIP_NO_WIDE_CHAIN=1; else unset JXL_HIP_NO_WIDE_CHAIN; fi
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-realistic --no-extras --no-cpu-baseline --main-tree-shape 1 --main-texture 5 --wide-first $1 --in-flight $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'wide_first': $1, 'in_flight': $2, 'chain': $3, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'steady': d.get('steady_state_ms_per_step'), 'stage_ms': d.get('stage_ms'), 'first_steps_ms': d['step_end_ms'][:4]}))"
done
