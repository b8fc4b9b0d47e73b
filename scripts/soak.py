"""One-off soak of the randomised parity tests with seeds beyond those in the suite.
usage: python scripts/soak.py [first_seed] [last_seed] [test names...]   (default 10 59, both randomised tests)"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(ROOT)
import torch
import test_gpu_parity as T

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 10
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 59
fails = []
for name in (sys.argv[3:] or ("test_random_decks_forward_loss_gradient", "test_one_sweep_kernel_random_geometry")):
    fn = getattr(T, name)
    for seed in range(lo, hi + 1):
        try:
            fn(torch, seed)
        except Exception as e:   # noqa: BLE001 -- report every failing seed
            fails.append((name, seed, repr(e)[:300]))
            traceback.print_exc()
        if seed % 10 == 0:
            print(name, "seed", seed, "failures so far", len(fails), flush=True)
print("soak: seeds %d..%d of 2 tests, %d failures" % (lo, hi, len(fails)))
for f in fails:
    print("  FAIL", f)
sys.exit(1 if fails else 0)
