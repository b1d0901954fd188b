#!/usr/bin/env python3
"""Times the 16-queries-per-pass VALU scan (f32 euclid / manhattan, SQ L1) at 10 M x 768: the paths without a matrix-core kernel."""
import ctypes as C, sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import qdrant_amd as qa
from qdrant_amd import _ffi as F
lib = F.lib()
dev = torch.device("cuda", 0)
n, dim, top = 10_000_000, 768, 10
rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
F.check(lib.qmx_synth_fill_f32(0, 7, 0, n, dim, F.ptr(rows)))
queries = torch.empty((16, dim), dtype=torch.float32, device=dev)
F.check(lib.qmx_synth_fill_f32(0, 8, 0, 16, dim, F.ptr(queries)))
torch.cuda.synchronize()
for dist in (qa.Distance.Euclid, qa.Distance.Manhattan):
    st = qa.VectorStorage(rows, dist)
    for Q in (8, 16):
        qh = C.c_void_p()
        F.check(lib.qmx_query_create(st._h, F.ptr(queries), Q, C.byref(qh)))
        F.check(lib.qmx_query_set_timing(qh, 1))
        out = torch.zeros((Q, top, 2), dtype=torch.int32, device=dev)
        cnt = torch.zeros((Q,), dtype=torch.int32, device=dev)
        ms, nl = C.c_float(), C.c_uint32()
        for _ in range(2):
            F.check(lib.qmx_search_topk_async(qh, top, None, 0, F.ptr(out), F.ptr(cnt)))
        F.check(lib.qmx_query_timing(qh, C.byref(ms), C.byref(nl)))
        for _ in range(5):
            F.check(lib.qmx_search_topk_async(qh, top, None, 0, F.ptr(out), F.ptr(cnt)))
        F.check(lib.qmx_query_timing(qh, C.byref(ms), C.byref(nl)))
        print(json.dumps({"distance": dist.name, "Q": Q, "scan_ms": round(ms.value / nl.value, 3), "qps": round(Q / (ms.value / nl.value) * 1e3, 1)}), flush=True)
        F.check(lib.qmx_query_destroy(qh))
    st.close()
