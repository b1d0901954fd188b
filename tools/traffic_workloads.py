#!/usr/bin/env python3
"""Launches, a few times each, every kernel that carries a `roofline` object in bench.py - at bench.py's sizes - so that a `rocprofv3 --pmc FETCH_SIZE`
/ `--pmc WRITE_SIZE` pass over this script yields their HBM traffic per launch (tools/pmc_traffic.sh collects and summarises; bench.py reads the
summary, profiles/pmc_traffic.json, and attaches `traffic` to a roofline object when the kernel symbol AND the workload shape match).

  c2   10 M x 768 f32 cosine: exact track at Q = 1, 8, 16, 32, 64; prefilter over the half copy at Q = 128 and 256
  c3   the latent rows as SQ-int8 (Q = 1, 32) and as TurboQuant 4-bit (Q = 1, 32)
  c4   10 M x 1536 PQ m = 96: prefilter at Q = 32, exact kernel at Q = 1
  hnsw SQ walk (d = 768) and PQ walk (d = 1536), ef = 128, --hnsw-queries searches per launch, over graphs of --hnsw-rows points built on the device
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--hnsw-rows", type=int, default=2_000_000)
    ap.add_argument("--hnsw-queries", type=int, default=8192)
    ap.add_argument("--what", default="c2,c3,c4,hnsw")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import qdrant_amd as qa
    from qdrant_amd import _ffi as F
    lib = F.lib()
    dev = torch.device("cuda", 0)
    n, top, reps = args.rows, 10, args.reps
    what = args.what.split(",")
    QROW0 = 1 << 40

    def latent(seed, row0, count, dim):
        x = torch.empty((count, dim), dtype=torch.float32, device=dev)
        F.check(lib.qmx_synth_fill_latent_f32(0, seed, row0, count, dim, 32, 1.0, F.ptr(x)))
        F.check(lib.qmx_preprocess_f32(0, int(qa.Distance.Cosine), F.ptr(x), count, dim, F.ptr(x)))
        return x

    def scans(storage, queries, Q, label, exact=False, k=top):
        qa.set_option("no_split_scan", 1 if exact else -1)
        try:
            qh = C.c_void_p()
            qb = queries[:Q].contiguous()
            F.check(lib.qmx_query_create(storage._h, F.ptr(qb), Q, C.byref(qh)))
            out = torch.zeros((Q, k, 2), dtype=torch.int32, device=dev)
            cnt = torch.zeros((Q,), dtype=torch.int32, device=dev)
            for _ in range(reps + 1):
                F.check(lib.qmx_search_topk_async(qh, k, None, 0, F.ptr(out), F.ptr(cnt)))
            F.check(lib.qmx_query_synchronize(qh))
            print(label, "Q=%d" % Q, F.last_kernel(qh), flush=True)
            lib.qmx_query_destroy(qh)
        finally:
            qa.set_option("no_split_scan", -1)

    if "c2" in what:
        dim = 768
        rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
        F.check(lib.qmx_synth_fill_f32(0, 0x5EED0002, 0, n, dim, F.ptr(rows)))
        F.check(lib.qmx_preprocess_f32(0, int(qa.Distance.Cosine), F.ptr(rows), n, dim, F.ptr(rows)))
        queries = torch.empty((256, dim), dtype=torch.float32, device=dev)
        F.check(lib.qmx_synth_fill_f32(0, 0x5EED0003, 0, 256, dim, F.ptr(queries)))
        st = qa.VectorStorage(rows, qa.Distance.Cosine, flags=F.SEG_HALF_COPY)
        for Q in (1, 8, 16, 32, 64):
            scans(st, queries, Q, "c2 exact", exact=True)
        for Q in (128, 256):
            scans(st, queries, Q, "c2 prefilter")
        st.close()
        del rows, st
        torch.cuda.empty_cache()
    if "c2i8" in what:           # the C2 block through its int8 copy (bench.py's default since the end of round 3)
        dim = 768
        rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
        F.check(lib.qmx_synth_fill_f32(0, 0x5EED0002, 0, n, dim, F.ptr(rows)))
        F.check(lib.qmx_preprocess_f32(0, int(qa.Distance.Cosine), F.ptr(rows), n, dim, F.ptr(rows)))
        queries = torch.empty((256, dim), dtype=torch.float32, device=dev)
        F.check(lib.qmx_synth_fill_f32(0, 0x5EED0003, 0, 256, dim, F.ptr(queries)))
        st = qa.VectorStorage(rows, qa.Distance.Cosine, flags=F.SEG_I8_COPY)
        scans(st, queries, 128, "c2 int8 prefilter")
        st.close()
        del rows, st
        torch.cuda.empty_cache()
    if "c3" in what:
        dim = 768
        rows = latent(0x5EED0003, 0, n, dim)
        queries = latent(0x5EED0003, QROW0, 64, dim)
        quant = qa.ScalarQuantizer.fit(rows, dim, qa.Distance.Dot)
        p = quant.params()
        codes = torch.empty((n, quant.quantized_vector_size()), dtype=torch.uint8, device=dev)
        F.check(lib.qmx_sq_encode(0, int(qa.Distance.Dot), C.byref(p), F.ptr(rows), n, dim, F.ptr(codes)))
        enc = qa.EncodedVectorsU8(codes, quant)
        del codes
        for Q in (1, 32):
            scans(enc, queries, Q, "c3 sq", k=2 * top)
        enc.close()
        tq = qa.TurboQuantizer(dim, qa.Distance.Dot, 0)
        tp = tq.params()
        tcodes = torch.empty((n, tq.quantized_vector_size()), dtype=torch.uint8, device=dev)
        F.check(lib.qmx_tq_encode(0, int(qa.Distance.Dot), dim, C.byref(tp), F.ptr(rows), n, F.ptr(tcodes)))
        tenc = qa.EncodedVectorsTQ(tcodes, tq)
        del tcodes
        for Q in (1, 32):
            scans(tenc, queries, Q, "tq4", k=2 * top)
        tenc.close()
        del rows
        torch.cuda.empty_cache()
    if "c4" in what:
        dim, chunk = 1536, 16
        rows = latent(0x5EED0004, 0, n, dim)
        queries = latent(0x5EED0004, QROW0, 64, dim)
        sample = rows[::max(1, n // 10000)][:10000].contiguous()
        cen = torch.zeros((256, dim), dtype=torch.float32, device=dev)
        import numpy as np
        iters = np.zeros(dim // chunk, dtype=np.uint32)
        F.check(lib.qmx_pq_train(0, F.ptr(sample), sample.shape[0], dim, chunk, 256, 100, 1e-5, 1, F.ptr(cen), F.ptr(iters)))
        quant = qa.ProductQuantizer(dim, qa.Distance.Dot, chunk, cen.cpu().numpy(), lut_mfma=True)
        p = quant.params()
        codes = torch.empty((n, quant.m), dtype=torch.uint8, device=dev)
        F.check(lib.qmx_pq_encode(0, C.byref(p), F.ptr(rows), n, dim, F.ptr(codes)))
        enc = qa.EncodedVectorsPQ(codes, quant)
        for Q in (32, 1):
            scans(enc, queries, Q, "c4 pq", k=2 * top)
        enc.close()
        del rows, codes
        torch.cuda.empty_cache()
    if "hnsw" in what:
        hn, nq = args.hnsw_rows, args.hnsw_queries
        for name, dim in (("sq", 768), ("pq", 1536)):
            rows = latent(0x5EED0003 if name == "sq" else 0x5EED0004, 0, hn, dim)
            queries = latent(0x5EED0003 if name == "sq" else 0x5EED0004, QROW0, nq, dim)
            vs = qa.VectorStorage(rows, qa.Distance.Cosine)
            if name == "sq":
                quant = qa.ScalarQuantizer.fit(rows, dim, qa.Distance.Dot)
                p = quant.params()
                codes = torch.empty((hn, quant.quantized_vector_size()), dtype=torch.uint8, device=dev)
                F.check(lib.qmx_sq_encode(0, int(qa.Distance.Dot), C.byref(p), F.ptr(rows), hn, dim, F.ptr(codes)))
                enc = qa.EncodedVectorsU8(codes, quant)
                graph = qa.GraphLayers.build(enc, m=16, ef_construct=100, seed=42)
            else:
                import numpy as np
                sample = rows[::max(1, hn // 10000)][:10000].contiguous()
                cen = torch.zeros((256, dim), dtype=torch.float32, device=dev)
                iters = np.zeros(dim // 16, dtype=np.uint32)
                F.check(lib.qmx_pq_train(0, F.ptr(sample), sample.shape[0], dim, 16, 256, 100, 1e-5, 1, F.ptr(cen), F.ptr(iters)))
                quant = qa.ProductQuantizer(dim, qa.Distance.Dot, 16, cen.cpu().numpy(), lut_mfma=True)
                p = quant.params()
                codes = torch.empty((hn, quant.m), dtype=torch.uint8, device=dev)
                F.check(lib.qmx_pq_encode(0, C.byref(p), F.ptr(rows), hn, dim, F.ptr(codes)))
                enc = qa.EncodedVectorsPQ(codes, quant)
                graph = qa.GraphLayers.build(enc, m=16, ef_construct=100, seed=42, original=vs)
            scorer = qa.new_raw_scorer(queries, enc)
            for _ in range(reps + 1):
                res, scored = graph.search(2 * top, 128, scorer, with_scored=True)
            print("hnsw", name, "rows=%d searches=%d scored/query=%.1f" % (hn, nq, scored / nq), F.last_kernel(scorer._h), flush=True)
            del scorer, graph, enc, vs, rows, codes
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
