python -m pytest tests/test_extractor_gpu.py -x -q -m gpu 2>&1 | tail -2
for st in 2 3 0; do
echo "FAST stop=$st: $(MSORB_FAST_DEBUG_STOP=$st python bench.py --steps 5 --warmup 2 --cpu-pairs 0 --isolated 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["stage_ms_per_step"]["fast"])')"
done
