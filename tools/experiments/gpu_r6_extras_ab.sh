#!/bin/bash
# round 6: the latency legs of bench.py (single frame, one_pass_128) with and without the 8K / JPEG legs in front of them
for extra in "--no-8k" ""; do
  python bench.py --gpus 1 --steps 6 --warmup 2 --no-realistic --no-cpu-baseline $extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'flags': '$extra', 'single': {k: v for k, v in d.get('single_frame_ms', {}).items() if k.endswith('_ms') or k == 'value'}, 'one_pass': {k: v for k, v in d.get('one_pass_128', {}).items() if k != 'what'},
 'cfg4': d['config'].get('workload_8k_modular_squeeze_u16', {}).get('value'), 'jpeg': d['config'].get('workload_jpeg_transcode_420', {}).get('value'), 'api': {k: v.get('mpixel_per_s') for k, v in d.get('api_concurrent', {}).items() if k.startswith('threads')}}))"
done
