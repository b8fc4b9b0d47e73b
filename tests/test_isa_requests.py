"""Static check of the hand-written vector-memory requests of the configs[3] sampler (project_rolling, k_form_factor_2d.inc).

The sampler requests the lines that enter its rolling window with `global_load` instructions written in inline assembly, one sample
ahead of their use, and waits for them with its own `s_waitcnt vmcnt(n)`.  The compiler does not know that a request's destination
registers hold nothing until the data has arrived: correctness rests on NO instruction touching them between the request and a wait
that covers it.  A first version violated that for odd table sizes (copies in front of a tail's wait) and failed intermittently; this
test reads the device assembly of the kernels that inline the walk and proves the property for every request in every loop:

    for each global_load inside a loop, walking forward (around the back edge once): before the first instruction that reads or writes
    one of its destination registers there is an `s_waitcnt vmcnt(n)` with n <= the number of vector-memory loads issued after it.

Needs hipcc (cross-compiles without a GPU), a few seconds per kernel."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

KERNELS = {
    "forward": "template __global__ void tsff::k_form_factor_2d<1, false, 1, false>(tsff::KStatic, const double*, const double*, int, double, double, int, long, long, double*, double*);",
    "adjoint": "template __global__ void tsff::k_form_factor_2d_adj<1, false, 1>(tsff::KStatic, const double*, const double*, int, double, double, int, long, long, const double*, double*, double*, const double*);",
    "forward_save": "template __global__ void tsff::k_form_factor_2d<1, false, 1, true>(tsff::KStatic, const double*, const double*, int, double, double, int, long, long, double*, double*);",
}


def _assembly(inst):
    with tempfile.TemporaryDirectory() as d:
        src, out = os.path.join(d, "one.hip"), os.path.join(d, "one.s")
        open(src, "w").write('#define TSFF_NO_API\n#include "%s"\n%s\n' % (os.path.join(ROOT, "tsadar_amd", "csrc", "tsff_kernels.hip"), inst))
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                        "-o", out, src], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def _regs(tok):
    """'v[8:11]' -> {8, 9, 10, 11}; 'v47' -> {47}; anything else -> empty"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _touched(args):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", args):
        out |= _regs(tok)
    return out


def _check(asm):
    lines = asm.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN4tsff\w+:", l)]
    checked = 0
    for s in starts:
        insts, labels = [], {}
        for l in lines[s + 1:]:
            if l.startswith(".Lfunc_end"):
                break
            t = l.split(";")[0].strip()
            if not t:
                continue
            m = re.match(r"^(\.LBB\w+):", t)
            if m:
                labels[m.group(1)] = len(insts)
                continue
            if t.startswith("."):
                continue
            p = t.split(None, 1)
            insts.append((p[0], p[1] if len(p) > 1 else ""))
        loops = [(labels[a.strip()], i) for i, (op, a) in enumerate(insts) if op.startswith("s_cbranch") and a.strip() in labels and labels[a.strip()] <= i]
        for a, b in loops:
            body = insts[a:b + 1]
            if any(op.startswith("s_cbranch") or op.startswith("s_branch") for op, _ in body[:-1]):
                continue   # (not an innermost straight-line loop)
            n = len(body)
            for k, (op, args) in enumerate(body):
                if not op.startswith("global_load") or not re.search(r",\s*s\[\d+:\d+\]", args):
                    continue   # (only the hand-written requests: SGPR base + 32-bit lane offset; the compiler waits for its own loads)
                dst = _regs(args.split(",")[0].strip())
                later_loads, covered = 0, False
                for step in range(1, 2 * n):
                    op2, args2 = body[(k + step) % n]
                    if op2.startswith("s_waitcnt"):
                        m = re.search(r"vmcnt\((\d+)\)", args2)
                        if m and int(m.group(1)) <= later_loads:
                            covered = True
                    if op2.startswith("global_load"):
                        if _regs(args2.split(",")[0].strip()) & dst:
                            assert covered, ("request overwritten before its data was waited for", k, (k + step) % n, op2, args2)
                            break
                        later_loads += 1
                        continue
                    if _touched(args2) & dst:
                        assert covered, ("a request's destination is touched before a wait covers it", op, args, "->", op2, args2)
                        break
                else:
                    raise AssertionError(("request never consumed", op, args))
                checked += 1
    return checked


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("name", sorted(KERNELS))
def test_request_registers_are_untouched_until_waited_for(name):
    n = _check(_assembly(KERNELS[name]))
    assert n >= 8 * 12, n   # eight walk forms (four in flight x two samples x six requests... at least twelve per loop)
