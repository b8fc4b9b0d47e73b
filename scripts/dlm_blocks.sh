#!/bin/bash
# bench.py --dlm over the number of column blocks of the pipelined step (TSFF_OPT_DLM_BLOCKS): 1 = one stream
for nb in 1 2 4 8 16; do
  python bench.py --dlm --dlm-blocks $nb --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('dlm blocks $nb | value', round(d['value']), 'ms/step', round(d['ms_per_step'],4))"
done
