#!/bin/bash
# A/B of an experimental libmsorb build against the in-tree one on ONE box, alternating runs:
#   gpurun -- 'bash tools/ab_lib.sh tools/_libmsorb_base.so [rounds]'
# prints, per run: value (Mkeypoints/s), ms_per_step, stage_ms_per_step (kernels alone on the GPU)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
ALT=$1; N=${2:-3}
line() { python bench.py --lean --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), d['ms_per_step'], d.get('stage_ms_per_step'))"; }
for i in $(seq $N); do
  echo "in-tree  $(line)"
  echo "alt      $(MSORB_LIB=$R/$ALT line)"
done
