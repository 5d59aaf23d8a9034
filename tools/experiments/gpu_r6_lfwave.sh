#!/bin/bash
# round 6: the wave-wide LF fast path — parity subset, then single-frame stage times with and without it
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lf or full_size_4k or sample or vardct or lane or simple" > gpurun_out/r06/lfwave_tests.txt 2>&1
tail -5 gpurun_out/r06/lfwave_tests.txt
python tools/experiments/gpu_r6_single.py > gpurun_out/r06/single_wave.txt 2>&1
JXL_HIP_NO_WAVE_LF=1 python tools/experiments/gpu_r6_single.py > gpurun_out/r06/single_nowave.txt 2>&1
cat gpurun_out/r06/single_wave.txt gpurun_out/r06/single_nowave.txt
