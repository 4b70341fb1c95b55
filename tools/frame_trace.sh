#!/bin/bash
# rocprofv3 kernel + memory-copy + HIP-runtime trace of tools/frame_chain.py: the LAST tracking frame (one msorb_track_frontend_motion
# + one msorb_search_local_points) as a merged timeline -> stdout; raw CSVs under gpurun_out/frame_trace/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/frame_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $OUT -o t -- python tools/frame_chain.py 40 > $OUT/run.log 2>&1 )
tail -1 $OUT/run.log
python $R/tools/frame_trace_print.py $OUT
