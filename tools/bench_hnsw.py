#!/usr/bin/env python3
"""HNSW build + search on the device (C3 / C4 style path of BASELINE.json), one JSON line per scorer.

  rows   : N x dim f32 generated ON THE DEVICE: `--data clustered` (mixture of 4096 gaussians around unit-norm centres,
           per-coordinate sigma 0.35 / sqrt(dim): low intrinsic dimension, recall is meaningful) or `--data iid` (N(0,1)
           per coordinate as hnsw_quantized_search_test.rs:66-78: in d = 768 nearest neighbours are nearly equidistant,
           recall collapses for the CPU reference and the device alike); cosine-normalised
  graph  : `--build gpu` qmx_hnsw_build (device, batch-parallel) or `--build cpu` the oracle's parallel builder
  scorer : f32 | sq (EncodedVectorsU8, dot over normalised rows) | pq (EncodedVectorsPQ chunk 16) | tq (EncodedVectorsTQ, 4 bits)
  build  : `--build-over f32|sq|tq`: through the original rows or through the quantized scorer (tq: qmx_hnsw_build_quantized with the original rows)
  search : qmx_hnsw_search of `--nq` queries in one launch (+ qmx_rescore with the f32 rows for sq / pq, oversampling 2)
  checks : first `--check` searches against the CPU oracle walking THE SAME graph (ids + score bits), recall@10 against
           exact brute force (device), CPU baseline = oracle search, one thread, on the same graph (rows <= --cpu-max-rows)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--scorer", default="sq", help="comma list of f32 | sq | pq | tq (TurboQuant 4 bits)")
    ap.add_argument("--build", choices=["gpu", "cpu"], default="gpu")
    ap.add_argument("--build-over", choices=["f32", "sq", "tq"], default="f32",
                    help="storage the device build scores through: the original vectors, or the SQ-int8 codes as the reference does "
                         "when the segment is quantized (hnsw/build.rs:334-341)")
    ap.add_argument("--data", choices=["clustered", "iid"], default="clustered")
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--ef-construct", type=int, default=100)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--top", type=int, default=10)
    ap.add_argument("--nq", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", type=int, default=64)
    ap.add_argument("--cpu-queries", type=int, default=256)
    ap.add_argument("--cpu-max-rows", type=int, default=2_000_000, help="above this the oracle (host copy of the rows) is skipped")
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--filter", type=float, default=0.0, help="also run a payload-filtered search (allow bitmap of this selectivity): plain filtered walk vs ACORN")
    args = ap.parse_args()

    import numpy as np
    import torch
    import qdrant_amd as qa
    from qdrant_amd import _ffi as F
    import oracle_ffi as O   # the checker and the CPU baseline

    lib = F.lib()
    dev = torch.device("cuda", 0)
    n, dim, nq, top = args.rows, args.dim, args.nq, args.top
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0003)

    def make(count, centres):
        out = torch.empty((count, dim), dtype=torch.float32, device=dev)
        for s in range(0, count, 1 << 20):
            e = min(count, s + (1 << 20))
            x = torch.randn((e - s, dim), generator=gen, device=dev, dtype=torch.float32)
            if centres is not None:
                idx = torch.randint(0, centres.shape[0], (e - s,), generator=gen, device=dev)
                x = centres[idx] + x * (0.35 / dim ** 0.5)
            out[s:e] = x
        F.check(lib.qmx_preprocess_f32(0, int(qa.Distance.Cosine), F.ptr(out), count, dim, F.ptr(out)))
        torch.cuda.synchronize()
        return out

    t0 = time.time()
    centres = None
    if args.data == "clustered":
        centres = torch.randn((4096, dim), generator=gen, device=dev, dtype=torch.float32)
        centres = centres / centres.norm(dim=1, keepdim=True)
    rows = make(n, centres)
    queries_d = make(nq, centres)
    queries = queries_d.cpu().numpy()
    t_data = time.time() - t0
    with_oracle = n <= args.cpu_max_rows
    host_rows = rows.cpu().numpy() if with_oracle else None
    st = O.DenseStorage(O.F32, O.COSINE, host_rows) if with_oracle else None

    sq_cache = {}

    def make_sq():
        """SQ-int8 twin of the rows (min / max interval), encoded on the device, as a device-resident EncodedVectorsU8."""
        if sq_cache:
            return sq_cache["quant"], sq_cache["enc"], sq_cache["codes"]
        mn, mx = float(rows.min().item()), float(rows.max().item())
        quant = qa.ScalarQuantizer(dim, qa.Distance.Dot, (np.float32(mx) - np.float32(mn)) / np.float32(127.0), np.float32(mn))
        p = quant.params()
        codes_d = torch.empty((n, quant.quantized_vector_size()), dtype=torch.uint8, device=dev)
        F.check(lib.qmx_sq_encode(0, int(qa.Distance.Dot), C.byref(p), F.ptr(rows), n, dim, F.ptr(codes_d)))
        enc = qa.EncodedVectorsU8.__new__(qa.EncodedVectorsU8)
        enc.quantizer, enc.distance, enc.datatype, enc.dim, enc.count, enc._keep, enc._sq = quant, quant.distance, None, dim, n, None, p
        d = F.SegmentDesc()
        d.dtype, d.distance, d.dim, d.n, d.data, d.device_id, d.sq = F.DTYPE_SQ_U8, int(qa.Distance.Dot), dim, n, F.ptr(codes_d).value, 0, C.pointer(p)
        enc._h = C.c_void_p()
        F.check(lib.qmx_segment_create(C.byref(d), C.byref(enc._h)))
        sq_cache.update(quant=quant, enc=enc, codes=codes_d.cpu().numpy() if with_oracle else None)
        del codes_d
        return sq_cache["quant"], sq_cache["enc"], sq_cache["codes"]

    tq_cache = {}

    def make_tq():
        """TurboQuant 4-bit twin of the rows, encoded on the device (qmx_tq_encode)."""
        if tq_cache:
            return tq_cache["quant"], tq_cache["enc"], tq_cache["codes"]
        quant = qa.TurboQuantizer(dim, qa.Distance.Dot, O.TQ_BITS4)
        p = quant.params()
        codes_d = torch.empty((n, quant.quantized_vector_size()), dtype=torch.uint8, device=dev)
        t0 = time.time()
        F.check(lib.qmx_tq_encode(0, int(qa.Distance.Dot), dim, C.byref(p), F.ptr(rows), n, F.ptr(codes_d)))
        torch.cuda.synchronize()
        tq_cache["encode_s"] = time.time() - t0
        host = codes_d.cpu().numpy()
        del codes_d
        tq_cache.update(quant=quant, enc=qa.EncodedVectorsTQ(host, quant), codes=host)
        return tq_cache["quant"], tq_cache["enc"], tq_cache["codes"]

    vs = qa.VectorStorage(rows, qa.Distance.Cosine)        # adopts the device block
    build_storage, original = vs, None
    if args.build == "gpu" and args.build_over == "sq":
        build_storage = make_sq()[1]
    if args.build == "gpu" and args.build_over == "tq":
        build_storage, original = make_tq()[1], vs
    t0 = time.time()
    if args.build == "gpu":
        graph = qa.GraphLayers.build(build_storage, m=args.m, ef_construct=args.ef_construct, seed=42, max_batch=args.max_batch, original=original)
        t_build = time.time() - t0
        plain = graph.export_plain()
        walker = O.Hnsw.from_plain(plain, n) if with_oracle else None
    else:
        assert with_oracle, "--build cpu needs the rows on the host"
        walker = O.Hnsw(st, m=args.m, ef_construct=args.ef_construct, seed=42, threads=args.threads)
        t_build = time.time() - t0
        plain = walker.export_plain()
        graph = qa.GraphLayers.from_plain(plain)
    n_links0 = int(plain.offsets[n]) if n else 0

    exact_n = min(256, nq)
    exact = qa.BatchFilteredSearcher(queries[:exact_n], vs, top).peek_top_all()
    raw_scorer = qa.new_raw_scorer(queries, vs)

    def recall(res):
        return sum(len(set(a["idx"].tolist()) & set(b["idx"].tolist())) for a, b in zip(res[:exact_n], exact)) / float(exact_n * top)

    for which in args.scorer.split(","):
        oracle_search = None
        oversample = 1
        if which == "f32":
            row_bytes, scorer = dim * 4, raw_scorer
            if with_oracle:
                oracle_search = lambda qs: walker.search_dense(st, qs, args.top, args.ef)                       # noqa: E731
        elif which == "sq":
            quant, enc, host_codes_sq = make_sq()
            row_bytes, oversample = quant.quantized_vector_size(), 2
            if with_oracle:
                osq = O.SqOracle(O.DOT, dim, quant.alpha, quant.offset)
                osq.rows = host_codes_sq
                oracle_search = lambda qs: walker.search_sq(st, osq, qs, args.top * 2, args.ef)                 # noqa: E731
            scorer = qa.new_raw_scorer(queries, enc)
        elif which == "tq":
            quant, enc, host_codes_tq = make_tq()
            row_bytes, oversample = quant.quantized_vector_size(), 2
            if with_oracle:
                otq = O.TqOracle(O.DOT, dim, O.TQ_BITS4)
                otq.rows = host_codes_tq
                oracle_search = lambda qs: walker.search_tq(st, otq, qs, args.top * 2, args.ef)                 # noqa: E731
            scorer = qa.new_raw_scorer(queries, enc)
        else:
            chunk = 16
            sample = rows[:10000].cpu().numpy()
            cen, _ = qa.pq_train(sample, dim, chunk, 256, max_iterations=100, accuracy=1e-5)
            quant = qa.ProductQuantizer(dim, qa.Distance.Dot, chunk, cen)
            p = quant.params()
            codes_d = torch.empty((n, quant.m), dtype=torch.uint8, device=dev)
            F.check(lib.qmx_pq_encode(0, C.byref(p), F.ptr(rows), n, dim, F.ptr(codes_d)))
            host_codes = codes_d.cpu().numpy()
            del codes_d
            enc = qa.EncodedVectorsPQ(host_codes, quant)
            row_bytes, oversample = quant.m, 2
            if with_oracle:
                opq = O.PqOracle(O.DOT, dim, chunk, cen)
                opq.codes = host_codes
                oracle_search = lambda qs: walker.search_pq(st, opq, qs, args.top * 2, args.ef)                 # noqa: E731
            scorer = qa.new_raw_scorer(queries, enc)
        stop = args.top * oversample
        F.check(lib.qmx_query_set_timing(scorer._h, 1))
        got = graph.search(stop, args.ef, scorer)           # warm-up (allocates the visited bitmaps)
        ms, nl = C.c_float(), C.c_uint32()
        F.check(lib.qmx_query_timing(scorer._h, C.byref(ms), C.byref(nl)))
        wall, scored = [], 0
        for _ in range(args.reps):
            t0 = time.time()
            got, scored = graph.search(stop, args.ef, scorer, with_scored=True)
            wall.append(time.time() - t0)
        F.check(lib.qmx_query_timing(scorer._h, C.byref(ms), C.byref(nl)))
        kernel_ms = ms.value / max(nl.value, 1)
        final = got
        t_rescore = 0.0
        if oversample > 1:                                   # postprocess_search_result: rescore with the original vectors
            ids = np.zeros((nq, stop), dtype=np.uint32)
            cnt = np.zeros(nq, dtype=np.uint32)
            for i, r in enumerate(got):
                ids[i, :len(r)] = r["idx"]
                cnt[i] = len(r)
            t0 = time.time()
            final = raw_scorer.rescore(ids, top, cnt)
            t_rescore = time.time() - t0
        out = {
            "metric": "HNSW search QPS (device-resident walk)", "scorer": which, "data": args.data, "build": args.build, "build_over": args.build_over if args.build == "gpu" else "f32",
            "rows": n, "dim": dim, "m": args.m, "ef_construct": args.ef_construct, "ef": args.ef, "top": top, "oversampling": oversample, "nq": nq,
            "build_s": round(t_build, 2), "build_points_per_s": round(n / max(t_build, 1e-9), 1), "links_level0": n_links0,
            "qps_kernel": round(nq / (kernel_ms * 1e-3), 1), "kernel_ms": round(kernel_ms, 3),
            "qps_wall_incl_copies": round(nq / min(wall), 1), "rescore_wall_s": round(t_rescore, 4),
            "points_scored_per_query": round(scored / nq, 1),
            "gather_GBps": round(scored * row_bytes / (kernel_ms * 1e-3) / 1e9, 1), "row_bytes": int(row_bytes),
            "recall_at_10": round(recall(final), 4), "data_s": round(t_data, 1),
        }
        if oracle_search is not None:
            want = oracle_search(queries[:args.check])      # rows and queries are already normalised: Cosine == Dot here
            out["oracle_walk_same_ids"] = "%d/%d" % (sum(int(a["idx"].tolist() == b["idx"].tolist()) for a, b in zip(got, want)), len(want))
            out["oracle_walk_same_score_bits"] = "%d/%d" % (
                sum(int(np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32))) for a, b in zip(got, want)), len(want))
            # every query whose id list differs: is it a tie (equal score bits at the differing positions, or at the cut after `top`)?
            diffs = []
            for qi, (a, b) in enumerate(zip(got, want)):
                if a["idx"].tolist() != b["idx"].tolist():
                    pos = [i for i in range(min(len(a), len(b))) if a["idx"][i] != b["idx"][i]]
                    ids = sorted(set(a["idx"][pos].tolist()) | set(b["idx"][pos].tolist()))
                    rs = scorer.score_points(ids)[qi] if hasattr(scorer, "score_points") else None
                    dup = None
                    if host_rows is not None and len(ids) == 2:
                        dup = bool(np.array_equal(host_rows[ids[0]], host_rows[ids[1]]))
                    diffs.append({"query": qi, "positions": pos, "device_ids": a["idx"][pos].tolist(), "oracle_ids": b["idx"][pos].tolist(),
                                  "device_score_bits": [int(x) for x in a["score"].view(np.uint32)[pos]], "oracle_score_bits": [int(x) for x in b["score"].view(np.uint32)[pos]],
                                  "ids_involved": ids, "their_scores_bits": None if rs is None else [int(x) for x in rs.view(np.uint32)],
                                  "rows_bit_identical": dup})
            out["oracle_walk_differences"] = diffs
            t0 = time.time()
            cpu_res = oracle_search(queries[:args.cpu_queries])
            t_cpu = time.time() - t0
            out["cpu_baseline"] = {"qps_one_thread": round(args.cpu_queries / t_cpu, 1), "kind": "port", "cores": 1, "host_cores": os.cpu_count()}
            if oversample == 1:
                out["cpu_recall_at_10"] = round(recall(cpu_res), 4)
        print(json.dumps(out), flush=True)

        if args.filter > 0.0 and which == "f32":
            # ScorerFilters' payload filter as an allow bitmap: the plain walk only traverses passing points, ACORN (graph_layers.rs:154-243)
            # explores through the others
            allowed = (torch.rand(n, generator=gen, device=dev) < args.filter).cpu().numpy()
            scorer.set_filter(allowed)
            fs = qa.BatchFilteredSearcher(queries[:exact_n], vs, top)
            fs.scorer.set_filter(allowed)
            exact_f = fs.peek_top_all()
            rec = lambda res: sum(len(set(a["idx"].tolist()) & set(b["idx"].tolist())) for a, b in zip(res[:exact_n], exact_f)) / float(max(1, sum(len(b) for b in exact_f)))   # noqa: E731
            for acorn in (False, True):
                graph.search(top, args.ef, scorer, acorn=acorn)
                F.check(lib.qmx_query_timing(scorer._h, C.byref(ms), C.byref(nl)))
                t0 = time.time()
                res, scored = graph.search(top, args.ef, scorer, with_scored=True, acorn=acorn)
                wall_f = time.time() - t0
                F.check(lib.qmx_query_timing(scorer._h, C.byref(ms), C.byref(nl)))
                kms = ms.value / max(nl.value, 1)
                print(json.dumps({"metric": "filtered HNSW search QPS (device-resident walk)", "algorithm": "acorn" if acorn else "hnsw", "scorer": which,
                                  "filter_selectivity": args.filter, "rows": n, "dim": dim, "m": args.m, "ef": args.ef, "top": top, "nq": nq,
                                  "qps_kernel": round(nq / (kms * 1e-3), 1), "kernel_ms": round(kms, 3), "qps_wall_incl_copies": round(nq / wall_f, 1),
                                  "points_scored_per_query": round(scored / nq, 1),
                                  "recall_at_10_vs_exact_filtered": round(rec(res), 4)}), flush=True)
            scorer.set_filter(None)


if __name__ == "__main__":
    main()
