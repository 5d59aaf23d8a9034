# FETCH_SIZE of the pixel kernels with and without the XCD-contiguous tile mapping (run under gpurun)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in on off; do
  if [ $v = off ]; then export JXL_HIP_NO_XCD_SWIZZLE=1; else unset JXL_HIP_NO_XCD_SWIZZLE; fi
  rm -rf /tmp/pmc_$v
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_$v -o f -- python $R/bench.py --batch 256 --steps 1 --warmup 1 --no-pipeline --no-cpu-baseline --no-extras --no-verify > /tmp/pmc_$v.log 2>&1 < /dev/null
  echo "== swizzle $v"; python $R/tools/pmc_sum.py /tmp/pmc_$v | grep -E "kernel|Fused|IdctTile.*4.*true|IdctTileKernel<4, true>"
done
