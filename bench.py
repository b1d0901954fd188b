#!/usr/bin/env python3
"""bench.py — QPS of brute-force top-10 search on the MI355X scorer (BASELINE.json metric).

Workload at N=1 = BASELINE.json configs[1] ("C2"): one segment of 10 M x d=768 f32, cosine,
brute-force exact top-10, resident in HBM.  A *step* = one pass of the hot path over one batch of
Q queries: Metric::preprocess of the batch (qmx_query_update) + one scan of the whole segment with
per-query top-k (qmx_search_topk_async = BatchFilteredSearcher::peek_top_iter) [+ for N>1 the RCCL
all-gather of the per-GPU top-k and the k-way merge].  1024 distinct queries are cycled in batches.

N>1 (torchrun, one rank per GPU) = configs[4] ("C5"): rank r holds its own 10 M-row segment
(seed + r), every rank scores the same query batch against its segment, the per-rank top-k lists
(Q x 10 x 8 B) are all-gathered over RCCL/xGMI and merged (BatchResultAggregator semantics).
Weak scaling: per-GPU work is fixed.  The counted unit is one (query, 10 M-row segment) search, so
value = N * Q * steps / time; at N=1 this is plain QPS on C2.  The collection-level QPS of the
N-segment collection (= value / N) is reported in config.collection_qps.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes of one scan launch
(rows x 3072 B) / the scan kernel's mean duration, measured with HIP-event pairs recorded on the
kernel's own stream inside the timed region (qmx_query_set_timing / qmx_query_timing).
`cpu_baseline` = the CPU oracle (AVX2+FMA restatement of the reference's scorer and its
peek_top_iter loop) timed on this box's host cores on a bounded sample of the same rows.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak (same guide: v_mfma_f32_16x16x4_f32 / 32x32x2, 64 FLOP/clk/SIMD)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=64, help="queries per scan (Q)")
    ap.add_argument("--top", type=int, default=10)
    ap.add_argument("--nqueries", type=int, default=1024)
    ap.add_argument("--cpu-rows", type=int, default=1_000_000, help="rows of the CPU-baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-hbm-point", action="store_true", help="skip the secondary Q=16 (HBM-bound) measurement of the same scan")
    ap.add_argument("--verify", type=int, default=1, help="check the first batch against the oracle on the CPU sample")
    ap.add_argument("--hnsw-rows", type=int, default=1_000_000,
                    help="rows of the secondary HNSW measurement (device build + SQ search + rescoring), 0 = skip")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import numpy as np
    import qdrant_amd as qa
    from qdrant_amd import _ffi as F

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    lib = F.lib()
    n, dim, Q, top = args.rows, args.dim, args.batch, args.top
    seed = 0x5EED0002  # 0x5EED0000 + config id

    # ---- the stored block: generated and normalised on device, adopted without copying ----
    rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
    F.check(lib.qmx_synth_fill_f32(local_rank, seed + 16 * rank, 0, n, dim, F.ptr(rows)))
    F.check(lib.qmx_preprocess_f32(local_rank, int(qa.Distance.Cosine), F.ptr(rows), n, dim, F.ptr(rows)))
    storage = qa.VectorStorage(rows, qa.Distance.Cosine, device_id=local_rank)

    nbatches = max(1, args.nqueries // Q)
    queries = torch.empty((nbatches * Q, dim), dtype=torch.float32, device=dev)
    F.check(lib.qmx_synth_fill_f32(local_rank, seed + 1, 0, nbatches * Q, dim, F.ptr(queries)))
    qbytes = Q * dim * 4

    stream = torch.cuda.Stream(dev)  # every kernel, the RCCL gather and the merge are ordered on this stream
    torch.cuda.set_stream(stream)
    from qdrant_amd import sharded
    backend = sharded.HipBackend(storage, Q, local_rank, stream)      # owns the qmx_query of this rank
    qh = backend.qh
    F.check(lib.qmx_query_set_timing(qh, 1))
    searcher = sharded.ShardedSearcher(backend, n, Q, top, device=dev)  # scan -> all-gather -> merge (world > 1)
    out, counts = searcher.out, searcher.counts

    def step(i):
        b = i % nbatches
        qb = queries[b * Q:(b + 1) * Q]
        if world > 1:
            searcher.search(qb)
        else:
            backend.local_topk(qb, top, out, counts)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    fence()
    ms0, l0 = C.c_float(), C.c_uint32()
    F.check(lib.qmx_query_timing(qh, C.byref(ms0), C.byref(l0)))  # drop warm-up launches

    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0

    kms, kl = C.c_float(), C.c_uint32()
    F.check(lib.qmx_query_timing(qh, C.byref(kms), C.byref(kl)))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        km = torch.tensor([kms.value / max(1, kl.value)], dtype=torch.float64, device=dev)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kernel_ms = float(km.item())
    else:
        kernel_ms = kms.value / max(1, kl.value)

    row_bytes = dim * 4
    alg_bytes = n * row_bytes  # per scan launch (3072 B/row at d=768), queries/outputs negligible
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    value = world * Q * args.steps / elapsed

    if world == 1:
        workload = "C2: 1 segment %s x d=%d f32 cosine, brute-force exact top-%d, batch Q=%d" % (_human(n), dim, top, Q)
    else:
        workload = ("C5: %d segments (one per GPU) x %s x d=%d f32 cosine, top-%d, batch Q=%d, RCCL all-gather + merge"
                    % (world, _human(n), dim, top, Q))
    result = {
        "metric": "QPS @ recall@10, brute-force, d=%d %s vecs, f32 cosine top-%d" % (dim, _human(n), top),
        "value": round(value, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "rows_per_gpu": n, "dim": dim, "batch": Q, "top": top, "distinct_queries": nbatches * Q,
                   "unit_of_value": "(query, 10M-row segment) searches per second; at n_gpus=1 this is plain QPS",
                   "collection_qps": round(Q * args.steps / elapsed, 2)},
        "roofline": _roofline(n, dim, Q, kernel_ms, alg_bytes, achieved, int(kl.value)),
    }

    if rank == 0 and world == 1 and Q != 16 and not args.no_hbm_point:
        # the HBM-bound operating point of the same scan (north_star: >= 70 % of the HBM roofline on C2): 16 queries per pass, where the
        # kernel is a pure stream of the stored block; outside the timed region, same rows, same measurement (HIP events on the kernel's stream)
        try:
            result["roofline_hbm_point_q16"] = hbm_point(16, storage, queries, n, dim, top, local_rank, stream, lib, F, sharded, torch)
        except Exception as e:
            result["roofline_hbm_point_q16"] = {"error": repr(e)[:300]}
    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(args, rows, queries, out, counts, n, dim, Q, top, lib, qh, F, qa, np, torch)
    if rank == 0 and world == 1 and args.hnsw_rows > 0:
        try:
            result["hnsw"] = hnsw_section(args, dev, dim, top, lib, F, qa, np, torch)
        except Exception as e:  # the headline line must survive a failure of the secondary measurement
            result["hnsw"] = {"error": repr(e)[:300]}
    if rank == 0:
        print(json.dumps(result), flush=True)
    backend.close()
    if world > 1:
        dist.destroy_process_group()


def hbm_point(Qh, storage, queries, n, dim, top, local_rank, stream, lib, F, sharded, torch):
    backend = sharded.HipBackend(storage, Qh, local_rank, stream)
    try:
        F.check(lib.qmx_query_set_timing(backend.qh, 1))
        out = torch.zeros((Qh, top, 2), dtype=torch.int32, device=queries.device)
        counts = torch.zeros((Qh,), dtype=torch.int32, device=queries.device)
        nb = max(1, queries.shape[0] // Qh)
        for i in range(3):
            backend.local_topk(queries[(i % nb) * Qh:(i % nb + 1) * Qh], top, out, counts)
        torch.cuda.synchronize()
        ms, nl = C.c_float(), C.c_uint32()
        F.check(lib.qmx_query_timing(backend.qh, C.byref(ms), C.byref(nl)))      # drop the warm-up launches
        steps = 30
        t0 = time.perf_counter()
        for i in range(steps):
            backend.local_topk(queries[(i % nb) * Qh:(i % nb + 1) * Qh], top, out, counts)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        F.check(lib.qmx_query_timing(backend.qh, C.byref(ms), C.byref(nl)))
        kernel_ms = ms.value / max(1, nl.value)
        alg = n * dim * 4
        gbps = alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        return {"batch": Qh, "kernel": _kernel_name(dim, Qh), "kernel_ms": round(kernel_ms, 4), "launches_timed": int(nl.value),
                "algorithmic_bytes_per_launch": alg, "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(gbps / HBM_PEAK_GBPS, 4), "qps": round(Qh * steps / wall, 1), "ms_per_step": round(wall / steps * 1e3, 4)}
    finally:
        backend.close()


def hnsw_section(args, dev, dim, top, lib, F, qa, np, torch):
    """Secondary measurement, outside the timed region (the metric names "brute-force + HNSW"): the C3-style path on
    `--hnsw-rows` clustered rows: SQ-int8 encode, device HNSW build through the SQ scorer (qmx_hnsw_build), SQ-int8 walk (qmx_hnsw_search, oversampling 2),
    rescoring with the f32 rows (qmx_rescore), recall@10 against the exact device search.  tools/bench_hnsw.py is the
    full tool (CPU-oracle walk parity, 10 M rows)."""
    n, nq, ef, m = args.hnsw_rows, 8192, 128, 16
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0003)
    centres = torch.randn((4096, dim), generator=gen, device=dev, dtype=torch.float32)
    centres = centres / centres.norm(dim=1, keepdim=True)

    def make(count):
        x = centres[torch.randint(0, 4096, (count,), generator=gen, device=dev)] + torch.randn((count, dim), generator=gen, device=dev) * (0.35 / dim ** 0.5)
        F.check(lib.qmx_preprocess_f32(dev.index or 0, int(qa.Distance.Cosine), F.ptr(x), count, dim, F.ptr(x)))
        return x
    rows, queries_d = make(n), make(nq)
    torch.cuda.synchronize(dev)
    queries = queries_d.cpu().numpy()
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    # the reference's order for a quantized segment: quantize, then build the graph THROUGH the quantized scorer
    # (hnsw/build.rs:334-341), then search with it and rescore with the original vectors
    mn, mx = float(rows.min().item()), float(rows.max().item())
    quant = qa.ScalarQuantizer(dim, qa.Distance.Dot, (np.float32(mx) - np.float32(mn)) / np.float32(127.0), np.float32(mn))
    p = quant.params()
    codes = torch.empty((n, quant.quantized_vector_size()), dtype=torch.uint8, device=dev)
    F.check(lib.qmx_sq_encode(dev.index or 0, int(qa.Distance.Dot), C.byref(p), F.ptr(rows), n, dim, F.ptr(codes)))
    d = F.SegmentDesc()
    d.dtype, d.distance, d.dim, d.n, d.data, d.device_id, d.sq = F.DTYPE_SQ_U8, int(qa.Distance.Dot), dim, n, F.ptr(codes).value, dev.index or 0, C.pointer(p)
    enc = qa.EncodedVectorsU8.__new__(qa.EncodedVectorsU8)
    enc.quantizer, enc.distance, enc.datatype, enc.dim, enc.count, enc._keep, enc._sq, enc._h = quant, quant.distance, None, dim, n, None, p, C.c_void_p()
    F.check(lib.qmx_segment_create(C.byref(d), C.byref(enc._h)))
    del codes
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    graph = qa.GraphLayers.build(enc, m=m, ef_construct=100, seed=42)
    t_build = time.perf_counter() - t0
    scorer = qa.new_raw_scorer(queries, enc)
    raw = qa.new_raw_scorer(queries, vs)
    F.check(lib.qmx_query_set_timing(scorer._h, 1))
    graph.search(2 * top, ef, scorer)
    ms, nl = C.c_float(), C.c_uint32()
    F.check(lib.qmx_query_timing(scorer._h, C.byref(ms), C.byref(nl)))
    t0 = time.perf_counter()
    got, scored = graph.search(2 * top, ef, scorer, with_scored=True)
    ids = np.zeros((nq, 2 * top), dtype=np.uint32)
    cnt = np.zeros(nq, dtype=np.uint32)
    for i, r in enumerate(got):
        ids[i, :len(r)] = r["idx"]
        cnt[i] = len(r)
    final = raw.rescore(ids, top, cnt)
    wall = time.perf_counter() - t0
    # the same three steps as ONE call (qmx_search_quantized: oversampled walk -> rescoring -> top, candidates stay in HBM)
    qa.search_quantized(scorer, raw, top, oversampling=2.0, rescore=True, graph=graph, hnsw_ef=ef)
    t0 = time.perf_counter()
    fused = qa.search_quantized(scorer, raw, top, oversampling=2.0, rescore=True, graph=graph, hnsw_ef=ef)
    wall_fused = time.perf_counter() - t0
    same = sum(int(np.array_equal(a, b)) for a, b in zip(fused, final))
    F.check(lib.qmx_query_timing(scorer._h, C.byref(ms), C.byref(nl)))
    exact = qa.BatchFilteredSearcher(queries[:256], vs, top).peek_top_all()
    recall = sum(len(set(a["idx"].tolist()) & set(b["idx"].tolist())) for a, b in zip(final[:256], exact)) / (256.0 * top)
    kernel_ms = ms.value / max(nl.value, 1)
    return {"workload": "C3-style: %s x d=%d clustered rows, SQ-int8 encode, device HNSW build through the SQ scorer (m=%d, ef_construct=100), SQ-int8 walk ef=%d, oversampling 2 + f32 rescoring, %d queries per launch"
                        % (_human(n), dim, m, ef, nq),
            "build_s": round(t_build, 2), "build_points_per_s": round(n / t_build, 1),
            "search_qps_kernel": round(nq / (kernel_ms * 1e-3), 1), "search_kernel_ms": round(kernel_ms, 3),
            "search_qps_wall_incl_host_copies_and_rescoring": round(nq / wall, 1),
            "search_qps_wall_one_call_walk_rescore_on_device": round(nq / wall_fused, 1), "one_call_lists_equal_three_step_lists": "%d/%d" % (same, nq),
            "points_scored_per_query": round(scored / nq, 1), "gather_GBps": round(scored * quant.quantized_vector_size() / (kernel_ms * 1e-3) / 1e9, 1),
            "recall_at_10_after_rescoring": round(recall, 4)}


def cpu_baseline(args, rows, queries, out, counts, n, dim, Q, top, lib, qh, F, qa, np, torch):
    """Times the oracle (checker, never the product) on a bounded sample; also verifies the GPU
    result of batch 0 on that sample (same rows, bit-identical generator)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    S = min(n, args.cpu_rows)
    host_rows = rows[:S].cpu().numpy()
    host_q = queries[:Q].cpu().numpy()
    ost = O.DenseStorage(O.F32, O.COSINE, host_rows)
    enc = ost.encode_queries(host_q)
    threads = os.cpu_count() or 1
    reps, t0 = 0, time.perf_counter()
    while True:
        res = ost.peek_top(enc, top, encoded=True, threads=threads)
        reps += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or reps >= 1000:
            break
    cpu_qps = Q * reps / el * (S / n)
    ok = None
    if args.verify:
        # GPU search restricted to the sampled rows (ids = 0..S) must return the oracle's ids and scores
        ids = torch.arange(S, dtype=torch.int32, device=rows.device)
        F.check(lib.qmx_query_update(qh, F.ptr(queries)))
        F.check(lib.qmx_search_topk_async(qh, top, F.ptr(ids), S, F.ptr(out), F.ptr(counts)))
        F.check(lib.qmx_query_synchronize(qh))
        g = out.cpu().numpy()
        gi = g[:, :, 0].view(np.uint32)
        gs = g[:, :, 1].copy().view(np.float32)
        ok = all(gi[i].tolist() == res[i]["idx"].tolist() and
                 np.allclose(gs[i], res[i]["score"], rtol=1e-5, atol=0) for i in range(Q))
        if not ok:
            print("PARITY FAILURE: GPU top-k differs from the oracle on the CPU sample", file=sys.stderr)
    return {"value": round(cpu_qps, 3), "unit": "queries/s", "cores": threads, "kind": "port",
            "sample": "oracle peek_top_iter (AVX2+FMA dot, 64-id chunks, heap of %d) over the first %d of %d rows, Q=%d, "
                      "%d threads on disjoint row ranges, %d scans in %.1f s; QPS scaled by %d/%d to the full segment"
                      % (top, S, n, Q, threads, reps, el, S, n),
            "gpu_matches_oracle_on_sample": ok}


def _roofline(n, dim, Q, kernel_ms, alg_bytes, achieved_gbps, launches):
    """The dominant kernel against BOTH ceilings; `bound` is the one it sits closer to.  Up to 16 queries per pass the scan is
    an HBM stream (every row byte read once: SURVEY 8d, 3072 B / row at d = 768); the 32- / 64-query passes of scan_mfma16.hip
    do 2 * dim flops per (row, query) on the f32 matrix cores and cross over to the MFMA ceiling."""
    per_pass = min(Q, _queries_per_pass(dim, Q))
    flops = 2.0 * n * dim * per_pass
    tflops = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    hbm_frac, mfma_frac = achieved_gbps / HBM_PEAK_GBPS, tflops / MFMA_F32_PEAK_TFLOPS
    common = {"traffic": _pmc_traffic(n, dim, Q), "kernel": _kernel_name(dim, Q), "kernel_ms": round(kernel_ms, 4), "launches_timed": launches,
              "queries_per_launch": per_pass, "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_flops_per_launch": flops,
              "hbm": {"achieved_GBps": round(achieved_gbps, 1), "peak_GBps": HBM_PEAK_GBPS, "frac": round(hbm_frac, 4)},
              "mfma_f32": {"achieved_TFLOPs": round(tflops, 2), "peak_TFLOPs": MFMA_F32_PEAK_TFLOPS, "frac": round(mfma_frac, 4)}}
    if mfma_frac > hbm_frac:
        return dict({"bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(mfma_frac, 4)}, **common)
    return dict({"bound": "hbm", "achieved": round(achieved_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(hbm_frac, 4)}, **common)


def _queries_per_pass(dim, Q):
    """api.hip search_enqueue: 64 per pass on the chain-major kernel (f32, dim 256 / 512 / 768, more than 32 queries), else 32."""
    if dim % 128 == 0 and 768 < dim <= 2048:
        return 32 if Q > 16 else 16
    return 64 if (Q > 32 and dim % 128 == 0 and dim <= 768) else 32


def _pmc_traffic(n, dim, Q):
    """HBM bytes per scan launch from a separate `rocprofv3 --pmc FETCH_SIZE` pass (committed under
    profiles/); null when no such pass exists for this shape."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
        return d.get("%dx%d_q%d" % (n, dim, Q))
    except Exception:
        return None


def _kernel_name(dim, Q):
    """The scan kernel a batch of Q queries runs (api.hip search_enqueue / launch_scan): <= 4 queries per pass stream through the
    VALU kernel, 8..16 through scan_mfma.hip (v_mfma_f32_4x4x1), 17.. through the chain-major scan_mfma16.hip (v_mfma_f32_16x16x4)
    when the rows are 256 / 512 / 768 floats."""
    if dim % 128 == 0 and 768 < dim <= 2048 and Q > 16:
        return "scan_f32_mfma16_kernel<KSTEPS=%d,NW=8,NT=2> (v_mfma_f32_16x16x4_f32, 32 queries per pass)" % (dim // 128)
    m16 = dim % 128 == 0 and dim <= 768
    if Q > 32 and m16:
        return "scan_f32_mfma16_kernel<KSTEPS=%d,NW=8,NT=4> (v_mfma_f32_16x16x4_f32, 64 queries per pass)" % (dim // 128)
    qt = _pow2(min(Q, 32))
    if qt == 16 and dim % 128 == 0 and dim <= 2048:
        return "scan_f32_mfma16_kernel<KSTEPS=%d,NW=4,NT=1> (v_mfma_f32_16x16x4_f32, 16 queries per pass)" % (dim // 128)
    if qt == 32 and m16:
        return "scan_f32_mfma16_kernel<KSTEPS=%d,NW=4,NT=2> (v_mfma_f32_16x16x4_f32, 32 queries per pass)" % (dim // 128)
    if qt >= 8:
        return "scan_f32_mfma_kernel<QW=%d,QSPLIT=%d> (v_mfma_f32_4x4x1, %d queries per pass)" % (min(qt, 16), max(1, qt // 16), qt)
    return "scan_kernel<RowF32<DOT>,QT=%d>" % qt


def _human(n):
    return ("%dM" % (n // 1_000_000)) if n % 1_000_000 == 0 else ("%dk" % (n // 1000)) if n % 1000 == 0 else str(n)


def _pow2(x):
    p = 1
    while p < x:
        p <<= 1
    return p


if __name__ == "__main__":
    main()
