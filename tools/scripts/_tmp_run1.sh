cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bj2
rm -rf $OUT; mkdir -p $OUT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $OUT/$tag -o p --output-format csv -- python $GRAFT_REPO_ROOT/tests/gpu_benchjxl.py 1 > /tmp/pmc_$tag.log 2>&1 < /dev/null
  tail -1 /tmp/pmc_$tag.log
done
