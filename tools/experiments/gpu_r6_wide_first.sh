#!/bin/bash
# round 6: how many LF stages of a cold pipeline should take the wave-wide kernel (it is 2x faster per stream now) — K = 20 value and the time the first step is done
for wf in ${WF_LIST:-0 1 2 3 4 6}; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-realistic --no-extras --no-cpu-baseline --wide-first $wf 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'wide_first': $wf, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'steady': d.get('steady_state_ms_per_step'), 'first_steps_ms': d['step_end_ms'][:6]}))"
done
