python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --cpu-pairs 0 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["stage_ms_per_step"], d["roofline"]["frac"])'
python tools/_lat.py | tail -2
