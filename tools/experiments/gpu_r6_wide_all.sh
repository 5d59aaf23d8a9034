#!/bin/bash
# round 6: the wave-wide LF kernel for EVERY job of the throughput pipeline (wide_first large) against the SIMT LF kernel, at several pipeline depths
mkdir -p gpurun_out/r06
for cfg in "4 11 11" "100000 11 11" "100000 4 4" "100000 3 3" "100000 6 6" "4 6 6"; do
  set -- $cfg
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-realistic --no-extras --no-cpu-baseline --wide-first $1 --in-flight $2 --lf-streams $3 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'wide_first': $1, 'in_flight': $2, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'steady': d.get('steady_state_ms_per_step'), 'stage_ms': d.get('stage_ms'), 'first_step_ms': d['step_end_ms'][0], 'device_gb': round(d['device_bytes'] / 1e9, 1)}))"
done
