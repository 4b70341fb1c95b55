#!/bin/bash
# whole-bench A/B of environment-selected kernel variants: tools/bench_variants.sh "ENV=a,ENV2=b" ...
# usage: tools/_bench_variants.sh "ENV1=a,ENV2=b" ...  -> value / ms_per_step / pyramid stage per variant (2 runs each)
for v in "$@"; do for rep in 1 2; do
  env $(echo $v | tr , ' ') python bench.py --steps 60 --warmup 5 --cpu-pairs 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done; done
