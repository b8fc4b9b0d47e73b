"""The C-ABI boundary without a GPU: libtsff.so loads, exports every symbol include/tsff.h
declares, the ctypes mirror of tsff_config has the C layout, and the product path fails loudly when
there is no HIP device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tsff.h")


@pytest.fixture(scope="module")
def lib():
    from tsadar_amd import build, _lib

    build.build()
    return _lib.load()


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tsff_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    from tsadar_amd import _lib

    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"libtsff.so does not export {n}"
    assert sorted(_lib.EXPORTS) == names, "ctypes binding and header disagree"
    assert lib.tsff_abi_version() == _lib.ABI_VERSION


def test_config_struct_layout_matches_c():
    """Compile a probe with gcc that prints sizeof/offsetof of tsff_config; compare with ctypes."""
    from tsadar_amd import _lib

    fields = [f[0] for f in _lib.TsffConfig._fields_]
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"tsff.h\"\nint main(){\n"
    prog += 'printf("%zu\\n", sizeof(tsff_config));\n'
    for f in fields:
        prog += f'printf("%zu\\n", offsetof(tsff_config, {f}));\n'
    prog += "return 0;}\n"
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "p.c"), os.path.join(td, "p")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    vals = [int(v) for v in out]
    assert vals[0] == C.sizeof(_lib.TsffConfig)
    for f, off in zip(fields, vals[1:]):
        assert getattr(_lib.TsffConfig, f).offset == off, f


def test_slot_constants_match_header():
    from tsadar_amd import _lib

    src = open(HEADER).read()
    enum = re.search(r"enum\s*\{\s*TSFF_P_TE.*?\};", src, re.S).group(0)
    got = dict(re.findall(r"(TSFF_P_[A-Z0-9_]+)\s*=\s*(\d+)", enum))
    want = dict(TSFF_P_TE=_lib.P_TE, TSFF_P_NE=_lib.P_NE, TSFF_P_M=_lib.P_M, TSFF_P_LAM=_lib.P_LAM, TSFF_P_AMP1=_lib.P_AMP1,
                TSFF_P_AMP2=_lib.P_AMP2, TSFF_P_AMP3=_lib.P_AMP3, TSFF_P_NE_GRADIENT=_lib.P_NE_GRADIENT,
                TSFF_P_TE_GRADIENT=_lib.P_TE_GRADIENT, TSFF_P_UD=_lib.P_UD, TSFF_P_VA=_lib.P_VA, TSFF_P_ION0=_lib.P_ION0)
    assert {k: int(v) for k, v in got.items()} == want
    assert int(re.search(r"#define TSFF_NXI2 (\d+)", src).group(1)) == _lib.NXI2
    assert int(re.search(r"#define TSFF_MAX_ION (\d+)", src).group(1)) == _lib.MAX_ION


def test_create_rejects_bad_config_and_reports_error(lib):
    from tsadar_amd import _lib

    c = _lib.TsffConfig()
    h = C.c_void_p()
    assert lib.tsff_create(C.byref(c), C.byref(h)) != 0  # abi_version 0
    assert b"ABI" in lib.tsff_last_error(None)
    c.abi_version = _lib.ABI_VERSION
    c.npts = 1000
    assert lib.tsff_create(C.byref(c), C.byref(h)) != 0
    assert b"1024" in lib.tsff_last_error(None)
    assert not h.value


def test_product_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import decks
    import util
    from tsadar_amd import _lib
    from tsadar_amd.engine import Engine

    with pytest.raises(_lib.TsffError, match="no CPU fallback"):
        Engine(decks.deck_fit(), util.sa_fit(1))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tsadar_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn
            assert "from oracle" not in src and "import oracle" not in src, fn
