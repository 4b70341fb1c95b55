#!/bin/bash
# Per-phase timestamps of the level-0 quadtree instance of a single stereo frame: builds a marks variant of the library
# (-DMSORB_QT_MARKS, not shipped) and runs one stereo frame a few times.  Run on the GPU box: tools/qt_marks.sh
# QT_VARIANTS="-DA=1;-DB=2": one run per ';'-separated set of extra compiler flags for quadtree_kernels.hip (default: one run, no flags)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R/ms-slam_amd/csrc
mkdir -p /tmp/qtm && for f in *.hip; do [ $f = quadtree_kernels.hip ] && continue; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMSORB_QT_MARKS -c $f -o /tmp/qtm/${f%.hip}.o & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -x c++ -c orb_host.cc -o /tmp/qtm/orb_host.o
wait
IFS=';' read -ra VARS <<< "${QT_VARIANTS:- }"
for v in "${VARS[@]}"; do
  cd $R/ms-slam_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMSORB_QT_MARKS $v -c quadtree_kernels.hip -o /tmp/qtm/quadtree_kernels.o || continue
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/qtm/libmsorb_marks.so /tmp/qtm/*.o -lpthread
  cd $R
  echo "=== variant: $v"
  MSORB_LIB=/tmp/qtm/libmsorb_marks.so MSORB_QT_DEBUG=3 python - <<'PY'
import sys, os
sys.path[:0] = ["ms-slam_amd"]
import msorb
from msorb import synth
cfg = synth.KITTI
L, R = synth.stereo_pair(0, cfg["rows"], cfg["cols"], texture=os.environ.get("QT_TEXTURE", "default"))
ex = msorb.ORBextractor(2000, 1.2, 8, 20, 7)
for i in range(3):
    print("--- frame", i, flush=True)
    ex.extract_stereo(L, R, 0.537, 386.1448)
ex.close()
PY
done
