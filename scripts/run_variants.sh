for v in "$@"; do cp tsadar_amd/libtsff_$v.so tsadar_amd/libtsff.so; python bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/bv_$v.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/bv_$v.json'));print('$v', d['value'],d['roofline']['kernel_avg_ms'])"; done
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
