for i in 1 2 3; do
  (cd ab_prev && python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('prev', d['roofline']['kernel_avg_ms'])")
  python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('new ', d['roofline']['kernel_avg_ms'])"
done
