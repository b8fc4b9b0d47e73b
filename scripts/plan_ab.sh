for i in 1 2 3; do
  for p in 0 1; do
    python bench.py --steps 10 --warmup 2 --cpu-sample 0 --plan $p 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('plan $p', round(d['roofline']['kernel_avg_ms'],4), round(d['ms_per_step'],4))"
  done
done
