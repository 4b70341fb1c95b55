#!/bin/bash
# The full measurement run of a round, on the GPU box:  gpurun --timeout 1500 -- 'bash tools/round_run.sh'
# then, back in the container:  python tools/collect_profiles.py round1   (copies the summaries into profiles/)
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err; tail -c 600 gpurun_out/bench_r1.json
tools/gpu_trace.sh prof_r1g > gpurun_out/trace_r1g.txt 2>&1; tail -25 gpurun_out/trace_r1g.txt
tools/gpu_pmc.sh pmc_fetch "FETCH_SIZE" "" > gpurun_out/pmc_fetch.txt 2>&1
tools/gpu_pmc.sh pmc_write "WRITE_SIZE" "" > gpurun_out/pmc_write.txt 2>&1
tools/gpu_pmc.sh pmc_valu "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE" "" > gpurun_out/pmc_valu.txt 2>&1
python tools/latency_bench.py > gpurun_out/latency_r1.json 2>gpurun_out/latency_r1.err; cat gpurun_out/latency_r1.json
python tools/bow_bench.py > gpurun_out/bow_bench.json 2>/dev/null; cat gpurun_out/bow_bench.json
g++ -O2 -std=c++17 tools/latency_pair.cc -Iinclude -Lms-slam_amd -lmsorb -lpthread -o /tmp/latency_pair 2>/dev/null && LD_LIBRARY_PATH=ms-slam_amd:/opt/rocm/lib /tmp/latency_pair 300 > gpurun_out/latency_pair.json; cat gpurun_out/latency_pair.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o /tmp/valu_ubench 2>/dev/null && /tmp/valu_ubench > gpurun_out/valu_ubench.txt; tail -3 gpurun_out/valu_ubench.txt
