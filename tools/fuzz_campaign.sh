#!/bin/bash
# The fuzz campaign of a round on the GPU box (every case against the CPU oracle, bit-exact):
#   gpurun --timeout 2400 -- 'bash tools/fuzz_campaign.sh 5'     -> gpurun_out/fuzz/roundN_fuzz.txt
#   a second argument shifts every seed (a second campaign of a round, on a later tree): ... 5 100 -> roundN_fuzz_s100.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; N=${1:-5}; O=gpurun_out/fuzz; mkdir -p $O
SH=${2:-0}; S=$((N * 1000 + 9000 + SH)); TAG=$([ $SH = 0 ] || echo _s$SH)
{
echo "Round-$N fuzz campaigns (GPU box, every case against the CPU oracle, bit-exact): tools/fuzz_params.py $((S+1)) 500, fuzz_geom.py $((S+2)) 300,"
echo "fuzz_stereo.py $((S+3)) 60 6, fuzz_track.py 200 $((S+4)), fuzz_visibility.py $((S+5)) 200"
echo "== params"; timeout 900 python tools/fuzz_params.py $((S+1)) 500 2>&1 | grep -v amdgpu.ids > $O/params.log; grep -c " OK" $O/params.log; tail -2 $O/params.log
echo "== geom"; timeout 600 python tools/fuzz_geom.py $((S+2)) 300 2>&1 | grep -v amdgpu.ids > $O/geom.log; grep -c " OK" $O/geom.log; tail -2 $O/geom.log
echo "== stereo6"; timeout 600 python tools/fuzz_stereo.py $((S+3)) 60 6 2>&1 | grep -v amdgpu.ids > $O/stereo.log; grep -c " OK" $O/stereo.log; tail -2 $O/stereo.log
echo "== track"; timeout 600 python tools/fuzz_track.py 200 $((S+4)) 2>&1 | grep -v amdgpu.ids > $O/track.log; tail -2 $O/track.log
echo "== vis"; timeout 600 python tools/fuzz_visibility.py $((S+5)) 200 2>&1 | grep -v amdgpu.ids > $O/vis.log; tail -1 $O/vis.log
} > $O/round${N}_fuzz$TAG.txt 2>&1
cat $O/round${N}_fuzz$TAG.txt
