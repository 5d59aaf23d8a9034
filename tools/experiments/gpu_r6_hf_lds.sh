#!/bin/bash
# round 6: a floor under the HF stage's LDS request (JXL_HIP_HF_LDS_MIN): above 80 KB no two HF workgroups share a CU — do the pixel kernels beside them gain more than the HF stage loses?
for floor in ${FLOORS:-0 84000 110000}; do
  export JXL_HIP_HF_LDS_MIN=$floor
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-realistic --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'hf_lds_floor': $floor, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'steady': d.get('steady_state_ms_per_step'), 'stage_ms': d['stage_ms'], 'first_steps_ms': d['step_end_ms'][:4]}))"
done
