#!/bin/bash
# round 6: the cold-start / small-job LF launches with the wide alias layout alone in LDS (40 instead of 57 KB per workgroup) against both layouts (JXL_HIP_LF_WIDE_BOTH)
for both in 0 1; do
  if [ $both = 1 ]; then export JXL_HIP_LF_WIDE_BOTH=1; else unset JXL_HIP_LF_WIDE_BOTH; fi
  for rep in 1 2; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-realistic --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'both_layouts': $both, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'steady': d.get('steady_state_ms_per_step'), 'first_steps_ms': d['step_end_ms'][:6]}))"
  done
  python tools/experiments/gpu_r6_latency_after_legs.py none 2>&1 | grep "^{" | head -1 | cut -c1-80
done
