#!/bin/bash
# rocprofv3 kernel trace of the production bench loop -> gpurun_out/timeline (start / end per kernel and queue)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --steps 12 --warmup 4 --lean > $OUT/bench.log 2>&1 )
ls $OUT
