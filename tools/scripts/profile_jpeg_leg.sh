# PMC traffic + stand-alone kernel durations of JPEG-transcode-shaped frames (YCbCr 4:2:0 4K; plain batch decodes of 64 frames, twice): IdctSubsampledTileKernel, OutputKernel's upsampling branch
set -x
TAG=${TAG:-r06e}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/jpeg_$C -o p -- python $R/tools/experiments/one_batch_decode.py jpeg 64 2 > $R/gpurun_out/$TAG/jpeg_$C.log 2>&1 < /dev/null
done
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/jpeg_stats -o p -- python $R/tools/experiments/one_batch_decode.py jpeg 64 2 > $R/gpurun_out/$TAG/jpeg_stats.log 2>&1 < /dev/null
find $R/gpurun_out/$TAG -name "*agent_info.csv" -delete
tail -2 $R/gpurun_out/$TAG/jpeg_stats.log
