#!/usr/bin/env python
"""Same-box A/B of the launch plans of tsff_loss_grad (TSFF_OPT_LAUNCH_PLAN bit mask: 1 never interleave, 2 two-sweep kernel,
4 no 3-per-CU forward, 8 no base-point exchange, 16 features alternating in runs of eight workgroups).
usage: python scripts/ab_plan.py [rounds] [plans, comma separated: default 0,2] [extra bench args...]"""
import json, subprocess, sys
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
plans = [int(p) for p in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 2]
extra = sys.argv[3:]
for r in range(rounds):
    for plan in plans:
        out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "3", "--cpu-sample", "0", "--plan", str(plan)] + extra,
                             capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print("plan", plan, "ms/step %.4f" % d["ms_per_step"], "kernel %.4f" % d["roofline"]["kernel_avg_ms"], "median %.4f" % d["roofline"]["kernel_median_ms"], flush=True)
        except Exception as e:
            print("plan", plan, "failed", e, out.stderr[-500:], flush=True)
