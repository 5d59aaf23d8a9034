#!/bin/bash
# round 6: wave-wide HF kernel — parity of small decodes (the scheduler's latency mode takes it for jobs of up to 8 frames), then single-frame stage times
python -m pytest tests -x -q -m gpu -k "not full_size and not config3 and not pipeline_streams" 2>&1 | tail -5
python tools/experiments/gpu_r6_single.py 2>&1 | grep -v "amdgpu.ids\|Exception\|Traceback\|File\|TypeError"
