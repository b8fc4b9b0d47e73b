python bench.py --steps 10 --warmup 2 --cpu-sample 0 "$@" > gpurun_out/bx.json 2> gpurun_out/bx.err || tail -5 gpurun_out/bx.err
python -c "import json;d=json.load(open('gpurun_out/bx.json'));print(d['metric']); print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'])"
