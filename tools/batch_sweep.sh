#!/bin/bash
# throughput by batch size B (stereo pairs per msorb_extract_batch call), one JSON line per B.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for cfg in "1 1 2000" "16 8 1600" "64 8 400" "128 8 200" "256 8 100" "1024 8 24"; do set -- $cfg
  python bench.py --pairs $1 --unique-pairs $2 --steps $3 --warmup 10 --cpu-pairs 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'pairs_per_batch': $1, 'Mkeypoints_per_s': d['value'], 'ms_per_batch': d['ms_per_step'], 'batches_in_flight': d['config'].get('batches_in_flight')}))"
done
