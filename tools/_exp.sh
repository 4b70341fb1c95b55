python -m pytest tests/test_extractor_gpu.py tests/test_dropin_cpp_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 10 --warmup 2 --cpu-pairs 0 --isolated 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["stage_ms_per_step"])'
