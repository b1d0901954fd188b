#!/usr/bin/env python3
"""Brute-force scans of BASELINE.json configs[2] (C3: 10 M x 768 SQ-int8, dot) and configs[3] (C4: 10 M x 1536
PQ m = 96) at full size on one MI355X.  bench.py stays the headline (C2); these are the parity-at-full-size and
GB/s numbers of the other single-GPU configs.

Everything is generated, quantized and scanned on the device through the C-ABI; the CPU oracle (checker) verifies
the first rows of the encoded block and the top-k restricted to a sample.  One JSON line per (config, batch).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timed_scans(lib, F, qh, top, out, counts, reps):
    ms, nl = C.c_float(), C.c_uint32()
    F.check(lib.qmx_search_topk_async(qh, top, None, 0, F.ptr(out), F.ptr(counts)))
    F.check(lib.qmx_query_synchronize(qh))
    F.check(lib.qmx_query_timing(qh, C.byref(ms), C.byref(nl)))
    t0 = time.perf_counter()
    for _ in range(reps):
        F.check(lib.qmx_search_topk_async(qh, top, None, 0, F.ptr(out), F.ptr(counts)))
    F.check(lib.qmx_query_synchronize(qh))
    wall = (time.perf_counter() - t0) / reps
    F.check(lib.qmx_query_timing(qh, C.byref(ms), C.byref(nl)))
    return ms.value / max(nl.value, 1), nl.value / reps, wall


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--configs", default="c3,c4")
    ap.add_argument("--batches", default="1,4,16")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--sample", type=int, default=200_000, help="rows of the oracle check")
    ap.add_argument("--c4-zero-query", type=int, default=-1, help="C4: make this query of the batch all zeros (a degenerate LUT: it falls back to the exact scan, alone)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import qdrant_amd as qa
    from qdrant_amd import _ffi as F
    import oracle_ffi as O

    lib = F.lib()
    dev = torch.device("cuda", 0)
    n, top = args.rows, 10
    S = min(args.sample, n)

    def make_rows(dim, seed):
        rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
        F.check(lib.qmx_synth_fill_f32(0, seed, 0, n, dim, F.ptr(rows)))
        F.check(lib.qmx_preprocess_f32(0, int(qa.Distance.Cosine), F.ptr(rows), n, dim, F.ptr(rows)))
        torch.cuda.synchronize()
        return rows

    def segment(dtype, dim, data, sq=None, pq=None, flags=0):
        d = F.SegmentDesc()
        d.dtype, d.distance, d.dim, d.flags, d.n = dtype, int(qa.Distance.Dot), dim, flags, n
        d.data, d.device_id = F.ptr(data), 0
        if sq is not None:
            d.sq = C.pointer(sq)
        if pq is not None:
            d.pq = C.pointer(pq)
        h = C.c_void_p()
        F.check(lib.qmx_segment_create(C.byref(d), C.byref(h)))
        return h

    def run(name, seg, dim, row_bytes, queries_pre, check):
        for Q in [int(x) for x in args.batches.split(",")]:
            assert len(queries_pre) >= Q, "the config generates %d queries" % len(queries_pre)
            q = torch.from_numpy(queries_pre[:Q].copy()).to(dev)
            qh = C.c_void_p()
            F.check(lib.qmx_query_create(seg, F.ptr(q), Q, C.byref(qh)))
            F.check(lib.qmx_query_set_timing(qh, 1))
            out = torch.zeros((Q, top, 2), dtype=torch.int32, device=dev)
            counts = torch.zeros((Q,), dtype=torch.int32, device=dev)
            kms, launches, wall = timed_scans(lib, F, qh, top, out, counts, args.reps)
            cnt = F.Counters()
            F.check(lib.qmx_query_last_counters(qh, C.byref(cnt)))
            ok = check(qh, Q, out, counts)
            F.check(lib.qmx_query_destroy(qh))
            alg = n * row_bytes
            print(json.dumps({
                "config": name, "rows": n, "dim": dim, "row_bytes": row_bytes, "batch": Q, "top": top,
                "scan_kernel_ms": round(kms, 3), "launches_per_search": launches, "ms_per_search_wall": round(wall * 1e3, 3),
                "qps": round(Q / wall, 1), "achieved_GBps": round(alg / (kms * 1e-3) / 1e9, 1),
                "frac_of_8TBps": round(alg / (kms * 1e-3) / 1e9 / 8000.0, 4), "algorithmic_bytes_per_scan": alg,
                "prefilter_queries": int(cnt.prefilter_queries), "fallback_queries": int(cnt.fallback_queries), "verified_rows": int(cnt.verified_rows), "prefilter_candidates": int(cnt.prefilter_candidates),
                "topk_on_sample_matches_oracle": ok}), flush=True)

    if "c3" in args.configs:
        dim = 768
        rows = make_rows(dim, 0x5EED0003)
        mn, mx = float(rows.min().item()), float(rows.max().item())
        quant = qa.ScalarQuantizer(dim, qa.Distance.Dot, (np.float32(mx) - np.float32(mn)) / np.float32(127.0), np.float32(mn))
        p = quant.params()
        enc_rows = torch.empty((n, quant.quantized_vector_size()), dtype=torch.uint8, device=dev)
        t0 = time.perf_counter()
        F.check(lib.qmx_sq_encode(0, int(qa.Distance.Dot), C.byref(p), F.ptr(rows), n, dim, F.ptr(enc_rows)))
        torch.cuda.synchronize()
        t_enc = time.perf_counter() - t0
        seg = segment(F.DTYPE_SQ_U8, dim, enc_rows, sq=p)
        host_rows = rows[:S].cpu().numpy()
        host_enc = enc_rows[:S].cpu().numpy()
        del rows, enc_rows
        torch.cuda.empty_cache()
        osq = O.SqOracle(O.DOT, dim, quant.alpha, quant.offset)
        want_rows = osq.encode_rows(host_rows[:2000])
        enc_ok = bool(np.array_equal(want_rows, host_enc[:2000]))
        osq.rows = host_enc
        queries = O.preprocess(O.COSINE, O.synth(0x5EED0013, 0, 64, dim))
        ids = torch.arange(S, dtype=torch.int32, device=dev)

        def check(qh, Q, out, counts):
            F.check(lib.qmx_search_topk_async(qh, top, F.ptr(ids), S, F.ptr(out), F.ptr(counts)))
            F.check(lib.qmx_query_synchronize(qh))
            g = out.cpu().numpy()
            gs = g[:, :, 1].copy().view(np.float32)
            sc = osq.score_points(queries[:min(Q, 2)], np.arange(S))
            return enc_ok and all(np.array_equal(np.sort(sc[i])[::-1][:top].view(np.uint32), gs[i].view(np.uint32)) for i in range(min(Q, 2)))
        print(json.dumps({"config": "C3 encode", "sq_encode_s": round(t_enc, 3), "rows": n, "encoded_rows_match_oracle_first_2000": enc_ok}), flush=True)
        run("C3: 10M x 768 SQ-int8 dot, brute-force top-10", seg, dim, quant.quantized_vector_size(), queries, check)
        F.check(lib.qmx_segment_destroy(seg))

    if "f16" in args.configs:
        dim = 768
        rows = make_rows(dim, 0x5EED0002)
        rows16 = torch.empty((n, dim), dtype=torch.float16, device=dev)
        F.check(lib.qmx_cast_f32(0, F.DTYPE_F16, F.ptr(rows), n * dim, F.ptr(rows16)))
        torch.cuda.synchronize()
        host16 = rows16[:S].cpu().numpy().view(np.uint16)
        del rows
        torch.cuda.empty_cache()
        d = F.SegmentDesc()
        d.dtype, d.distance, d.dim, d.flags, d.n, d.data, d.device_id = F.DTYPE_F16, int(qa.Distance.Cosine), dim, F.SEG_DATA_ON_DEVICE, n, F.ptr(rows16).value, 0
        seg = C.c_void_p()
        F.check(lib.qmx_segment_create(C.byref(d), C.byref(seg)))
        ost = O.DenseStorage(O.F16, O.COSINE, host16)
        queries = O.synth(0x5EED0012, 0, 64, dim)
        ids = torch.arange(S, dtype=torch.int32, device=dev)

        def check(qh, Q, out, counts):
            F.check(lib.qmx_search_topk_async(qh, top, F.ptr(ids), S, F.ptr(out), F.ptr(counts)))
            F.check(lib.qmx_query_synchronize(qh))
            g = out.cpu().numpy()
            gs = g[:, :, 1].copy().view(np.float32)
            want = ost.peek_top(queries[:min(Q, 2)], top)
            return all(np.allclose(gs[i], want[i]["score"], rtol=1e-4, atol=1e-6) for i in range(min(Q, 2)))
        run("C2-f16: 10M x 768 f16 cosine, brute-force top-10", seg, dim, dim * 2, queries, check)
        F.check(lib.qmx_segment_destroy(seg))
        del rows16
        torch.cuda.empty_cache()

    if "tq" in args.configs:
        dim = 768
        for bits, label in ((O.TQ_BITS4, "4-bit"), (O.TQ_BITS2, "2-bit"), (O.TQ_BITS1, "1-bit")):
            rows = make_rows(dim, 0x5EED0007)
            quant = qa.TurboQuantizer(dim, qa.Distance.Dot, bits)
            p = quant.params()
            rb = quant.quantized_vector_size()
            enc_rows = torch.empty((n, rb), dtype=torch.uint8, device=dev)
            t0 = time.perf_counter()
            F.check(lib.qmx_tq_encode(0, int(qa.Distance.Dot), dim, C.byref(p), F.ptr(rows), n, F.ptr(enc_rows)))
            torch.cuda.synchronize()
            t_enc = time.perf_counter() - t0
            host_rows = rows[:S].cpu().numpy()
            host_enc = enc_rows[:S].cpu().numpy()
            del rows
            torch.cuda.empty_cache()
            d = F.SegmentDesc()
            d.dtype, d.distance, d.dim, d.flags, d.n, d.data, d.device_id = F.DTYPE_TQ, int(qa.Distance.Dot), dim, 0, n, F.ptr(enc_rows).value, 0
            d.tq = C.pointer(p)
            seg = C.c_void_p()
            F.check(lib.qmx_segment_create(C.byref(d), C.byref(seg)))
            del enc_rows
            torch.cuda.empty_cache()
            otq = O.TqOracle(O.DOT, dim, bits)
            enc_ok = bool(np.array_equal(otq.encode_rows(host_rows[:500]), host_enc[:500]))
            otq.rows = host_enc
            queries = O.preprocess(O.COSINE, O.synth(0x5EED0017, 0, 64, dim))
            S_tq = min(S, 20000)
            ids = torch.arange(S_tq, dtype=torch.int32, device=dev)

            def check_tq(qh, Q, out, counts):
                F.check(lib.qmx_search_topk_async(qh, top, F.ptr(ids), S_tq, F.ptr(out), F.ptr(counts)))
                F.check(lib.qmx_query_synchronize(qh))
                gs = out.cpu().numpy()[:, :, 1].copy().view(np.float32)
                sc = otq.score_points(queries[:1], np.arange(S_tq))
                return enc_ok and bool(np.array_equal(np.sort(sc[0])[::-1][:top].view(np.uint32), gs[0].view(np.uint32)))
            print(json.dumps({"config": "TQ %s encode" % label, "tq_encode_s": round(t_enc, 3), "rows": n,
                              "encoded_rows_match_oracle_first_500": enc_ok}), flush=True)
            run("TQ %s: 10M x %d TurboQuant dot, brute-force top-10" % (label, dim), seg, dim, rb, queries, check_tq)
            F.check(lib.qmx_segment_destroy(seg))

    if "bq" in args.configs:
        for dim in (768, 1536):
            rows = make_rows(dim, 0x5EED0005)
            rb = (dim + 127) // 128 * 16
            enc_rows = torch.empty((n, rb), dtype=torch.uint8, device=dev)
            t0 = time.perf_counter()
            F.check(lib.qmx_bq_encode(0, F.ptr(rows), n, dim, F.ptr(enc_rows)))
            torch.cuda.synchronize()
            t_enc = time.perf_counter() - t0
            host_rows = rows[:S].cpu().numpy()
            host_enc = enc_rows[:S].cpu().numpy()
            del rows
            torch.cuda.empty_cache()
            d = F.SegmentDesc()
            d.dtype, d.distance, d.dim, d.flags, d.n, d.data, d.device_id = F.DTYPE_BQ, int(qa.Distance.Cosine), dim, F.SEG_DATA_ON_DEVICE, n, F.ptr(enc_rows).value, 0
            seg = C.c_void_p()
            F.check(lib.qmx_segment_create(C.byref(d), C.byref(seg)))
            obq = O.BqOracle(O.COSINE, dim)
            enc_ok = bool(np.array_equal(obq.encode(host_rows[:2000]), host_enc[:2000]))
            obq.rows = host_enc
            queries = O.preprocess(O.COSINE, O.synth(0x5EED0015, 0, 64, dim))
            ids = torch.arange(S, dtype=torch.int32, device=dev)

            def check(qh, Q, out, counts):
                F.check(lib.qmx_search_topk_async(qh, top, F.ptr(ids), S, F.ptr(out), F.ptr(counts)))
                F.check(lib.qmx_query_synchronize(qh))
                g = out.cpu().numpy()
                gs = g[:, :, 1].copy().view(np.float32)
                sc = obq.score_points(queries[:1], np.arange(min(S, 20000)))
                full = out.cpu().numpy()
                del full
                return enc_ok and bool(np.all(np.diff(gs[0]) <= 0)) and float(gs[0][0]) >= float(sc[0].max())
            print(json.dumps({"config": "BQ encode d=%d" % dim, "bq_encode_s": round(t_enc, 3), "rows": n, "encoded_rows_match_oracle_first_2000": enc_ok}), flush=True)
            run("BQ: 10M x %d 1-bit (u128) cosine, brute-force top-10" % dim, seg, dim, rb, queries, check)
            F.check(lib.qmx_segment_destroy(seg))
            # the same rows against QueryEncoding::Scalar8bits queries (asymmetric quantization: 8 bit planes per query value)
            bqp = F.BqParams()
            bqp.encoding, bqp.query_encoding = F.BQ_ONE_BIT, F.BQ_QUERY_SCALAR_8BITS
            d.bq = C.pointer(bqp)
            seg = C.c_void_p()
            F.check(lib.qmx_segment_create(C.byref(d), C.byref(seg)))

            def check8(qh, Q, out, counts):
                F.check(lib.qmx_search_topk_async(qh, top, F.ptr(ids), S, F.ptr(out), F.ptr(counts)))
                F.check(lib.qmx_query_synchronize(qh))
                g = out.cpu().numpy()
                gs = g[:, :, 1].copy().view(np.float32)
                sc = obq.score_points_scalar(queries[:1], np.arange(min(S, 20000)), 8)
                return enc_ok and bool(np.all(np.diff(gs[0]) <= 0)) and float(gs[0][0]) >= float(sc[0].max())
            run("BQ: 10M x %d 1-bit rows, Scalar8bits queries, cosine, brute-force top-10" % dim, seg, dim, rb, queries, check8)
            F.check(lib.qmx_segment_destroy(seg))
            del enc_rows
            torch.cuda.empty_cache()

    if "c4" in args.configs:
        dim, chunk = 1536, 16
        rows = make_rows(dim, 0x5EED0004)
        host_rows = rows[:S].cpu().numpy()
        t0 = time.perf_counter()
        cen, kiters = qa.pq_train(host_rows[:10000], dim, chunk, 256, max_iterations=100, accuracy=1e-5)   # KMEANS_SAMPLE_SIZE / MAX_ITERATIONS / ACCURACY
        t_train = time.perf_counter() - t0
        quant = qa.ProductQuantizer(dim, qa.Distance.Dot, chunk, cen)
        p = quant.params()
        codes = torch.empty((n, quant.m), dtype=torch.uint8, device=dev)
        t0 = time.perf_counter()
        F.check(lib.qmx_pq_encode(0, C.byref(p), F.ptr(rows), n, dim, F.ptr(codes)))
        torch.cuda.synchronize()
        t_enc = time.perf_counter() - t0
        seg = segment(F.DTYPE_PQ, dim, codes, pq=p)
        host_codes = codes[:S].cpu().numpy()
        del rows, codes
        torch.cuda.empty_cache()
        opq = O.PqOracle(O.DOT, dim, chunk, cen)
        enc_ok = bool(np.array_equal(opq.encode(host_rows[:2000]), host_codes[:2000]))
        opq.codes = host_codes
        queries = O.preprocess(O.COSINE, O.synth(0x5EED0014, 0, max(64, max(int(x) for x in args.batches.split(","))), dim))
        if args.c4_zero_query >= 0:      # a degenerate LUT: that query alone takes the exact scan behind the prefilter (what does the straggler cost the batch?)
            queries[args.c4_zero_query] = 0.0
        ids = torch.arange(S, dtype=torch.int32, device=dev)

        def check(qh, Q, out, counts):
            F.check(lib.qmx_search_topk_async(qh, top, F.ptr(ids), S, F.ptr(out), F.ptr(counts)))
            F.check(lib.qmx_query_synchronize(qh))
            g = out.cpu().numpy()
            gs = g[:, :, 1].copy().view(np.float32)
            sc = opq.score_points(queries[:1], np.arange(S))
            return enc_ok and bool(np.array_equal(np.sort(sc[0])[::-1][:top].view(np.uint32), gs[0].view(np.uint32)))
        print(json.dumps({"config": "C4 train + encode", "pq_kmeans_train_s": round(t_train, 3), "kmeans_iterations_max": int(kiters.max()),
                          "pq_encode_s": round(t_enc, 3), "rows": n, "codes_match_oracle_first_2000": enc_ok}), flush=True)
        run("C4: 10M x 1536 PQ m=96 dot, brute-force top-10", seg, dim, quant.m, queries, check)
        F.check(lib.qmx_segment_destroy(seg))


if __name__ == "__main__":
    main()
