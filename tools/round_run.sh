#!/bin/bash
# The full measurement run of a round, on the GPU box:  gpurun --timeout 2400 -- 'bash tools/round_run.sh'
# then, back in the container:  python tools/collect_profiles.py round5   (copies the summaries into profiles/)
# Every profiler pass runs under its own timeout: a counter set the hardware refuses leaves rocprofv3 waiting forever.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/round; rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
python bench.py --unique-pairs 128 --steps 100 --warmup 10 --cpu-pairs 0 > $O/bench_unique128.json 2>/dev/null; tail -c 300 $O/bench_unique128.json
python bench.py --steps 100 --warmup 10 --cpu-pairs 0 > $O/bench_unique8_same_steps.json 2>/dev/null
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_gpu.txt
timeout -k 5 300 tools/gpu_trace.sh round/trace > $O/trace.txt 2>&1; tail -30 $O/trace.txt
PMC_TIMEOUT=300 tools/pmc_run.sh round/pmc_fetch "FETCH_SIZE" "" -- python bench.py --steps 2 --warmup 1 --lean --isolated > $O/pmc_fetch.txt 2>&1
PMC_TIMEOUT=300 tools/pmc_run.sh round/pmc_write "WRITE_SIZE" "" -- python bench.py --steps 2 --warmup 1 --lean --isolated > $O/pmc_write.txt 2>&1
PMC_TIMEOUT=300 tools/pmc_run.sh round/pmc_valu "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE" "" -- python bench.py --steps 2 --warmup 1 --lean --isolated > $O/pmc_valu.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/_fetch_calib 2>/dev/null; tools/fetch_calib.sh > $O/fetch_calib.txt 2>&1; cat $O/fetch_calib.txt
python tools/latency_bench.py > $O/latency.json 2> $O/latency.err; cat $O/latency.json
python tools/bow_bench.py > $O/bow_bench.json 2>/dev/null; cat $O/bow_bench.json
{ python tools/hamming_bench.py; python tools/hamming_bench.py --popcount; } > $O/hamming_bench.txt 2>&1; tail -5 $O/hamming_bench.txt
python tools/frame_chain.py 300 > $O/frame_chain.json 2>/dev/null; cat $O/frame_chain.json
tools/frame_trace.sh > $O/frame_trace.txt 2>&1; tail -45 $O/frame_trace.txt
tools/qt_marks.sh 2>&1 | tail -26 > $O/qt_marks.txt; tail -26 $O/qt_marks.txt
tools/batch_sweep.sh > $O/batch_sweep.jsonl 2>/dev/null; cat $O/batch_sweep.jsonl
g++ -O2 -std=c++17 tools/latency_pair.cc -Iinclude -Lms-slam_amd -lmsorb -lpthread -o /tmp/latency_pair 2>/dev/null && LD_LIBRARY_PATH=ms-slam_amd:/opt/rocm/lib /tmp/latency_pair 300 > $O/latency_pair.json; cat $O/latency_pair.json
g++ -O2 -std=c++17 -Itests/cv_stub -Ims-slam_amd/host -Iinclude tools/latency_class.cc ms-slam_amd/host/ORBextractor.cc -Lms-slam_amd -lmsorb -lpthread -o /tmp/latency_class 2>$O/latency_class.err && {
  LD_LIBRARY_PATH=ms-slam_amd:/opt/rocm/lib /tmp/latency_class 300 > $O/latency_class_on.json
  MSORB_HOST_PYRAMID=0 LD_LIBRARY_PATH=ms-slam_amd:/opt/rocm/lib /tmp/latency_class 300 > $O/latency_class_off.json; cat $O/latency_class_on.json $O/latency_class_off.json; }
g++ -O2 -std=c++17 tools/kf_store_bench.cc -Iinclude -Lms-slam_amd -lmsorb -lpthread -o /tmp/kf_store_bench 2>$O/kf_store.err && LD_LIBRARY_PATH=ms-slam_amd:/opt/rocm/lib /tmp/kf_store_bench > $O/kf_store_bench.json; cat $O/kf_store_bench.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o /tmp/valu_ubench 2>/dev/null && /tmp/valu_ubench > $O/valu_ubench.txt; tail -3 $O/valu_ubench.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_ubench2.hip -o /tmp/valu_ubench2 2>/dev/null && /tmp/valu_ubench2 > $O/valu_ubench2.txt; head -3 $O/valu_ubench2.txt
g++ -O2 -std=c++17 tools/visibility_bench.cc -Iinclude -Lms-slam_amd -lmsorb -o /tmp/visibility_bench 2>/dev/null && LD_LIBRARY_PATH=ms-slam_amd:/opt/rocm/lib /tmp/visibility_bench 300 > $O/visibility_bench.json; cat $O/visibility_bench.json
g++ -O2 -std=c++17 tools/latency_pair.cc -Iinclude -Lms-slam_amd -lmsorb -lpthread -o tools/_latency_pair 2>/dev/null; tools/track_trace.sh > $O/track_trace.txt 2>&1; tail -30 $O/track_trace.txt
tools/timeline.sh > /dev/null 2>&1; python tools/timeline_summary.py gpurun_out/timeline > $O/timeline_pipelined.txt 2>&1; tail -2 $O/timeline_pipelined.txt
# round 5: describe_kernel's L2 -> L1 fills and wait cycles, per-frame copies by kernel against SDMA
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  PMC_TIMEOUT=200 tools/pmc_run.sh round/dpmc_$n "$set" "describe" -- python bench.py --steps 2 --warmup 1 --lean --isolated 2>&1 | tail -2
done > $O/describe_pmc.txt 2>&1; cat $O/describe_pmc.txt
{ for i in 1 2; do python tools/per_frame_ab.py | tail -1; MSORB_FRAME_COPIES=sdma python tools/per_frame_ab.py | tail -1; done; } > $O/frame_copies_ab.txt 2>&1; cat $O/frame_copies_ab.txt
# round 5: what the frame's quadtree by quadrant path and the one-launch pyramid buy (the same process start to end per line: ms one image, ms stereo frame)
{ for i in 1 2; do echo "default           $(python tools/per_frame_ab.py | tail -1)"; echo "MSORB_QT_PATHS=0  $(MSORB_QT_PATHS=0 python tools/per_frame_ab.py | tail -1)"; echo "MSORB_PYR_TOWER=0 $(MSORB_PYR_TOWER=0 python tools/per_frame_ab.py | tail -1)"; echo "both off          $(MSORB_QT_PATHS=0 MSORB_PYR_TOWER=0 python tools/per_frame_ab.py | tail -1)"; done; } > $O/frame_paths_tower_ab.txt 2>&1; cat $O/frame_paths_tower_ab.txt
{ for c in 4 0 5 4 0; do echo "MSORB_QT_BATCH_PATHS=$c $(MSORB_QT_BATCH_PATHS=$c python bench.py --lean 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'))")"; done; } > $O/batch_paths_ab.txt 2>&1; cat $O/batch_paths_ab.txt
# per-frame latency on the three input classes (ms one image, two images, stereo frame, keypoints), with and without the path form
{ echo "default build:"; python tools/per_frame_classes.py 2>/dev/null | tail -4; echo "MSORB_QT_PATHS=0:"; MSORB_QT_PATHS=0 python tools/per_frame_classes.py 2>/dev/null | tail -4; } > $O/per_frame_classes.txt 2>&1; cat $O/per_frame_classes.txt
# round 5 (late): fast_cells_kernel's LDS pipe beside its VALU (the arc phase is bound by both), the descriptor stage's processing order
{ PMC_TIMEOUT=200 tools/pmc_run.sh round/fpmc_lds "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "fast_cells" -- python bench.py --steps 2 --warmup 1 --lean --isolated 2>&1 | tail -1
  PMC_TIMEOUT=200 tools/pmc_run.sh round/fpmc_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "fast_cells" -- python bench.py --steps 2 --warmup 1 --lean --isolated 2>&1 | tail -1; } > $O/fast_pmc.txt 2>&1; cat $O/fast_pmc.txt
{ for i in 1 2 3; do for o in 1 0; do echo "MSORB_DESC_ORDER=$o $(MSORB_DESC_ORDER=$o python bench.py --lean --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'))")"; done; done; } > $O/describe_order_ab.txt 2>&1; cat $O/describe_order_ab.txt
