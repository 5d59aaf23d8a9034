#!/bin/bash
# round 6: prepare workers of the pipeline (each parses + prepares + uploads one job and enqueues its LF stage) against the fill of a cold pipeline, K = 20
for pt in ${PT_LIST:-3 6 4}; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-realistic --no-extras --no-cpu-baseline --prepare-threads $pt 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'prepare_threads': $pt, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'steady': d.get('steady_state_ms_per_step'), 'first_steps_ms': d['step_end_ms'][:5], 'cores_busy': d['config'].get('host_cpu', {}).get('cores_busy')}))"
done
