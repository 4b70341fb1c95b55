#!/bin/bash
# FAST phase costs: bench --isolated with the kernel cut after phase 1..4 (MSORB_FAST_DEBUG_STOP; results invalid), for PMC / timing deltas
for s in 1 2 3 4 0; do
  MSORB_FAST_DEBUG_STOP=$s python bench.py --steps 3 --warmup 1 --cpu-pairs 0 --isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stop=$s fast_ms', d['stage_ms_per_step']['fast'])"
done
