"""The C-ABI library loads on a CPU-only box and exports every symbol include/qdrant_amd.h
declares; without a GPU every entry point fails loudly (no CPU fallback).  CPU only."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "qdrant_amd.h")).read()
    return sorted(set(re.findall(r"QMX_API\s+[\w\s\*]+?\b(qmx_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    from qdrant_amd import _ffi
    assert _declared() == sorted(_ffi.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from qdrant_amd import _ffi
    lib = _ffi.lib()  # raises if the .so is missing or a symbol is absent
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.qmx_abi_version() == 1


def test_struct_layouts_match_the_header():
    from qdrant_amd import _ffi
    assert C.sizeof(_ffi.ScoredPoint) == 8          # ScoredPointOffset is 8 bytes, #[repr(C)]
    assert C.sizeof(_ffi.Counters) == 32
    assert C.sizeof(_ffi.SqParams) == 20
    assert C.sizeof(_ffi.SegmentDesc) == 64


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import qdrant_amd as qa
    rows = np.zeros((4, 32), dtype=np.float32)
    with pytest.raises(qa.QmxError) as e:
        qa.VectorStorage(rows, qa.Distance.Dot)
    assert e.value.status == qa._ffi.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_product_code_never_touches_the_oracle():
    # oracle/ is test infrastructure: nothing under qdrant_amd/ may import, link or name it
    for base, _, files in os.walk(os.path.join(ROOT, "qdrant_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(base, f)).read()
                assert "qdrant_oracle" not in text and "oracle_ffi" not in text and "libqdrant_oracle" not in text, f
