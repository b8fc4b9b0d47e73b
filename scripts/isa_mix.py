#!/usr/bin/env python
"""Static instruction mix of one kernel from the device assembly (hipcc -S --cuda-device-only).

    python scripts/isa_mix.py tsff.s '<mangled kernel name prefix>' [--loops]

Splits the kernel into basic blocks, finds the loops (backward branches) and prints, per loop body and for the
whole kernel, the count of instructions by class: FP64 arithmetic (fma / mul / add), FP64 special (rcp, rsq,
ldexp, rndne, frexp, cvt, min/max, cmp), moves/selects (v_mov, v_cndmask, v_readfirstlane, v_accvgpr), integer
VALU, LDS, VMEM/SMEM, SALU, waitcnt/branch.  Used for the round's breakdown of k_spectrum's two sweeps."""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64", "v_pk_fma_f64", "v_pk_mul_f64", "v_pk_add_f64")):
        return "f64_arith"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
        return "f64_trans"
    if op.startswith(("v_ldexp_f64", "v_rndne_f64", "v_frexp", "v_cvt_", "v_trunc_f64", "v_floor_f64", "v_fract_f64", "v_ceil_f64")):
        return "f64_cvt"
    if op.startswith(("v_min_f64", "v_max_f64", "v_cmp", "v_cmpx", "v_div_", "v_trig")) or ("_class_" in op):
        return "f64_cmp_minmax" if "f64" in op else "int_cmp"
    if op.startswith(("v_cndmask", "v_mov", "v_accvgpr", "v_readfirstlane", "v_readlane", "v_writelane", "v_swap", "v_perm", "v_bfi", "v_pk_mov")):
        return "move_select"
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "int_valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer") or op.startswith("s_store"):
        return "smem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_branch", "s_cbranch", "s_endpgm", "s_sleep", "s_setprio")):
        return "ctrl"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    want_loops = "--loops" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.rstrip().split(";")[0].strip().endswith(":"))
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l)
    # instruction list with label positions
    insts, labels = [], {}
    for l in body:
        s = l.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB[\w]+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        insts.append((parts[0], parts[1] if len(parts) > 1 else ""))
    total = collections.Counter(classify(op) for op, _ in insts)
    print("kernel: %d instructions" % len(insts))
    for k, v in total.most_common():
        print("  %-16s %6d  %5.1f %%" % (k, v, 100.0 * v / len(insts)))
    if not want_loops:
        return
    loops = []
    for i, (op, arg) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")):
            t = arg.strip()
            if t in labels and labels[t] <= i:
                loops.append((labels[t], i))
    # innermost loops first; print each with its mix
    loops.sort(key=lambda ab: ab[1] - ab[0])
    print("\nloops (start, end, size):")
    for a, b in loops:
        n = b - a + 1
        if n < 40:
            continue
        c = collections.Counter(classify(op) for op, _ in insts[a:b + 1])
        valu = sum(v for k, v in c.items() if k in ("f64_arith", "f64_trans", "f64_cvt", "f64_cmp_minmax", "int_cmp", "move_select", "int_valu", "mfma"))
        print("  [%5d, %5d] %5d instr, %5d VALU: " % (a, b, n, valu) + ", ".join("%s %d" % kv for kv in c.most_common()))
        if "--ops" in sys.argv:
            oc = collections.Counter(op for op, _ in insts[a:b + 1])
            print("      " + ", ".join("%s %d" % kv for kv in oc.most_common(40)))


if __name__ == "__main__":
    main()
