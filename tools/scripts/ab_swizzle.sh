cd /root/repo
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('swizzle   ', d['value'], d['ms_per_step'], d['stage_ms'])"
JXL_HIP_NO_XCD_SWIZZLE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no swizzle', d['value'], d['ms_per_step'], d['stage_ms'])"
