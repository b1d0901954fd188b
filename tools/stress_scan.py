#!/usr/bin/env python3
"""Race screen of the chain-major scan (LDS-DMA ring, counted vmcnt waits): the same search repeated many times on 10 M x 768 must
return the identical lists every time, and the lists of the 4x4x1 / VALU kernels (QMX_NO_MFMA16=1).  Prints one line per case."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import qdrant_amd as qa
    from qdrant_amd import _ffi as F
    lib = F.lib()
    dev = torch.device("cuda", 0)
    n, dim, top, reps = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("DIM", 768)), 10, int(os.environ.get("REPS", 40))
    rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
    F.check(lib.qmx_synth_fill_f32(0, 0x5EED0002, 0, n, dim, F.ptr(rows)))
    F.check(lib.qmx_preprocess_f32(0, int(qa.Distance.Cosine), F.ptr(rows), n, dim, F.ptr(rows)))
    torch.cuda.synchronize()
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    ids = torch.randperm(n, device=dev)[: n // 3].to(torch.int32)
    bad = 0
    for Q in (16, 32, 64):
        q = torch.randn((Q, dim), device=dev, dtype=torch.float32)
        qh = C.c_void_p()
        F.check(lib.qmx_query_create(vs._h, F.ptr(q), Q, C.byref(qh)))
        for use_ids in (False, True):
            out = torch.zeros((Q, top, 2), dtype=torch.int32, device=dev)
            counts = torch.zeros((Q,), dtype=torch.int32, device=dev)

            def run():
                F.check(lib.qmx_search_topk_async(qh, top, F.ptr(ids) if use_ids else None, ids.numel() if use_ids else 0, F.ptr(out), F.ptr(counts)))
                F.check(lib.qmx_query_synchronize(qh))
                return out.clone()
            qa.set_option("no_mfma16", 1)
            ref = run()
            qa.set_option("no_mfma16", -1)
            diff = sum(int(not torch.equal(run(), ref)) for _ in range(reps))
            bad += diff
            print("Q=%d ids=%s: %d / %d runs differ from the reference kernels' lists" % (Q, use_ids, diff, reps), flush=True)
        F.check(lib.qmx_query_destroy(qh))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
