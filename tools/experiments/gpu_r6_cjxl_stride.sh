#!/bin/bash
# round 6: LF-group streams per SIMT wavefront (64 / --lane-stride-lf; the weighted-predictor instantiation spends four lanes per stream) on the cjxl-shaped and the headline frames, K = 20
for shape in 1 0; do
for ls in ${LS_LIST:-4 8 16 32}; do
  tex=$([ $shape = 1 ] && echo 5 || echo 0)
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-realistic --no-extras --no-cpu-baseline --main-tree-shape $shape --main-texture $tex --lane-stride-lf $ls 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'tree_shape': $shape, 'lane_stride_lf': $ls, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'steady': d.get('steady_state_ms_per_step'), 'stage_ms': d.get('stage_ms'), 'first_steps_ms': d['step_end_ms'][:3]}))"
done; done
