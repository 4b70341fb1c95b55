#!/bin/bash
# pyr_tower_kernel with 1024 / 512 / 256 threads per tile: stereo frame time of each build (GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R/ms-slam_amd/csrc
for t in 1024 512 256; do
  mkdir -p /tmp/tw$t
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMSORB_TOWER_THREADS=$t -c orb_kernels.hip -o /tmp/tw$t/orb_kernels.o &
done; wait
for t in 1024 512 256; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/tw$t/libmsorb.so /tmp/tw$t/orb_kernels.o $(ls *.o | grep -v orb_kernels.o) -lpthread
done
cd $R
for rep in 1 2; do for t in 1024 512 256; do echo "threads $t: $(MSORB_LIB=/tmp/tw$t/libmsorb.so python tools/per_frame_ab.py | tail -1)"; done; done
