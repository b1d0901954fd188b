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
    assert lib.qmx_abi_version() == 8


def test_struct_layouts_match_the_header():
    from qdrant_amd import _ffi
    assert C.sizeof(_ffi.ScoredPoint) == 8          # ScoredPointOffset is 8 bytes, #[repr(C)]
    assert C.sizeof(_ffi.Counters) == 56
    assert C.sizeof(_ffi.SqParams) == 20
    assert C.sizeof(_ffi.SegmentDesc) == 80          # + the qmx_tq_params pointer (ABI 3)
    assert C.sizeof(_ffi.TqParams) == 32
    assert C.sizeof(_ffi.BqParams) == 24


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


def test_plain_links_file_header_is_validated_before_any_device_work():
    """qmx_hnsw_create_from_plain_file parses graph_links/header.rs:9-20 on the host: malformed files are refused
    with BAD_ARG / NOT_SUPPORTED even on a box without a GPU; a well-formed one then needs the device."""
    import torch
    import oracle_ffi as O
    from qdrant_amd import _ffi as F
    rows = O.preprocess(O.COSINE, O.synth(5, 0, 200, 16))
    g = O.Hnsw(O.DenseStorage(O.F32, O.COSINE, rows), m=4, ef_construct=16, seed=1)
    p = g.export_plain()
    data = O.plain_links_file(p)
    assert len(data) % 8 == 0 and np.frombuffer(data[:32], dtype="<u8").tolist() == [200, len(p.level_offsets) - 1, len(p.neighbors), len(p.offsets)]

    def create(buf):
        d = F.HnswDesc()
        d.m, d.m0 = p.m, p.m0
        ep, epl = np.ascontiguousarray(p.ep_ids, dtype=np.uint32), np.ascontiguousarray(p.ep_levels, dtype=np.uint32)
        d.entry_point_ids, d.entry_point_levels, d.n_entry_points = ep.ctypes.data, epl.ctypes.data, len(ep)
        h = C.c_void_p()
        arr = np.frombuffer(buf, dtype=np.uint8)
        rc = F.lib().qmx_hnsw_create_from_plain_file(F.ptr(arr), len(arr), C.byref(d), C.byref(h))
        if rc == F.OK:
            F.lib().qmx_hnsw_destroy(h)
        return rc
    assert create(data[:40]) == F.ERR_BAD_ARG                                    # shorter than the header
    assert create(data[:-8]) == F.ERR_BAD_ARG                                    # truncated offsets section
    compressed = bytearray(data)
    compressed[8:16] = np.array([0xFFFFFFFFFFFFFF01], dtype="<u8").tobytes()     # HEADER_VERSION_COMPRESSED sits where levels_count is
    assert create(bytes(compressed)) == F.ERR_NOT_SUPPORTED
    bad = bytearray(data)
    off_neigh = 64 + 8 * (len(p.level_offsets) - 1) + 4 * 200
    bad[off_neigh:off_neigh + 4] = np.array([5000], dtype="<u4").tobytes()       # a link past point_count
    assert create(bytes(bad)) == F.ERR_OUT_OF_BOUNDS
    assert create(data) == (F.OK if torch.cuda.is_available() else F.ERR_NO_DEVICE)
