#!/bin/bash
# every bench variant once (value, ms/step, kernel ms) -- numbers quoted in DESIGN.md / README.md
run() { python bench.py --steps 10 --warmup 2 --cpu-sample 0 "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$*', '| value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel ms', round(d['roofline']['kernel_avg_ms'],4))"; }
run
run --forward-only
run --forward-only --batch 256
run --batch 256
run --batch 32768
run --ppp 2
run --ppp 5 --batch 1024
run --dlm
run --free-form
run --nvx 320
run --nvx 320 --dlm
run --forward-only --ppp 5 --batch 1024
run --forward-only --ppp 2
run --forward-only --ppp 5 --batch 16
run --ppp 5 --batch 16
