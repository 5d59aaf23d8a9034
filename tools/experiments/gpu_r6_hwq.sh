#!/bin/bash
# round 6: do the pipeline's streams share hardware queues?  (main + 11 LF + 2 HF + copy streams + the caller's own = 16 or more against GPU_MAX_HW_QUEUES=16)  K = 20
run() {
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-realistic --no-extras --no-cpu-baseline $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'cfg': '$1 $2', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'steady': d.get('steady_state_ms_per_step'), 'first_steps_ms': d['step_end_ms'][:5]}))"
}
GPU_MAX_HW_QUEUES=16 run "hwq16" ""
GPU_MAX_HW_QUEUES=24 run "hwq24" ""
GPU_MAX_HW_QUEUES=32 run "hwq32" ""
GPU_MAX_HW_QUEUES=16 run "hwq16" "--lf-streams 8"
GPU_MAX_HW_QUEUES=24 run "hwq24" "--lf-streams 8"
