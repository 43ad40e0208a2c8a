"""C-ABI boundary: libsvdx.so loads, exports every symbol include/svdx.h declares, and the product refuses to run
without a GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "svdx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svdx_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from svd_xtend_amd import build, kernels
    path = build.build()
    return kernels.load_library(path)


def test_header_and_binding_agree():
    from svd_xtend_amd import kernels
    assert header_symbols() == sorted(kernels.EXPORTED_SYMBOLS)


def header_prototypes():
    """name -> list of 'p' / 'i' / 'f' / 'l' per argument, parsed from the header's prototypes."""
    src = open(os.path.join(ROOT, "include", "svdx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(svdx_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        args = [a.strip() for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        kinds = []
        for a in args:
            if "*" in a:
                kinds.append("p")
            elif re.match(r"(const\s+)?float\b", a):
                kinds.append("f")
            elif re.match(r"(const\s+)?double\b", a):
                kinds.append("d")
            elif re.match(r"(const\s+)?(long|int64_t|size_t)\b", a):
                kinds.append("l")
            else:
                kinds.append("i")
        out[m.group(1)] = "".join(kinds)
    return out


def test_binding_signatures_match_header_prototypes():
    """A ctypes table that drifts from the header corrupts the call frame silently: compare them argument by argument."""
    from svd_xtend_amd import kernels
    protos = header_prototypes()
    for name, sig in kernels._SIGS.items():
        assert name in protos, name
        norm = sig.replace("z", "l")          # the table spells size_t 'z' and long 'l': same register class
        assert norm == protos[name], f"{name}: binding {sig} vs header {protos[name]}"


def test_library_exports_every_declared_symbol(lib):
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.svdx_version() >= 100


def test_error_reporting_without_device(lib):
    import ctypes
    # argument validation happens before any launch, so it can be exercised without a GPU
    rc = lib.svdx_gemm(None, None, None, 0, 0, 0, 0, 0, 0, None, None, 0, 0, 0, None, 0, None, None, 0, 1.0, 1, 0, 0, None, None, 0,
                       0, None)
    assert rc != 0
    buf = ctypes.create_string_buffer(256)
    lib.svdx_last_error(buf, 256)
    assert b"svdx_gemm" in buf.value


def test_launch_plan_handles_without_device(lib):
    """svdx_plan_*: recording state is per thread, an empty plan is a valid plan, a second begin is an error (no launch involved)."""
    import ctypes
    lib.svdx_plan_end.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    h = ctypes.c_void_p()
    assert lib.svdx_plan_end(ctypes.byref(h)) != 0               # nothing is being recorded
    assert lib.svdx_plan_begin() == 0
    assert lib.svdx_plan_begin() != 0
    assert lib.svdx_plan_end(ctypes.byref(h)) == 0 and h.value
    assert lib.svdx_plan_launches(h) == 0 and lib.svdx_plan_bytes(h) == 0
    assert lib.svdx_plan_replay(h, None) == 0                    # no launches: nothing to issue
    assert lib.svdx_plan_free(h) == 0
    assert lib.svdx_plan_replay(None, None) != 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only box check")
def test_product_fails_loudly_without_gpu():
    from svd_xtend_amd import kernels
    prev = kernels._backend
    kernels._backend = None
    try:
        with pytest.raises(kernels.SvdxError):
            kernels.backend()
    finally:
        kernels._backend = prev
