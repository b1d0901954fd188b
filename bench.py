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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=16, help="queries per scan (Q)")
    ap.add_argument("--top", type=int, default=10)
    ap.add_argument("--nqueries", type=int, default=1024)
    ap.add_argument("--cpu-rows", type=int, default=1_000_000, help="rows of the CPU-baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--verify", type=int, default=1, help="check the first batch against the oracle on the CPU sample")
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
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": _pmc_traffic(n, dim, Q),
                     "kernel": _kernel_name(Q), "kernel_ms": round(kernel_ms, 4),
                     "launches_timed": int(kl.value), "algorithmic_bytes_per_launch": alg_bytes},
    }

    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(args, rows, queries, out, counts, n, dim, Q, top, lib, qh, F, qa, np, torch)
    if rank == 0:
        print(json.dumps(result), flush=True)
    backend.close()
    if world > 1:
        dist.destroy_process_group()


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


def _pmc_traffic(n, dim, Q):
    """HBM bytes per scan launch from a separate `rocprofv3 --pmc FETCH_SIZE` pass (committed under
    profiles/); null when no such pass exists for this shape."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
        return d.get("%dx%d_q%d" % (n, dim, Q))
    except Exception:
        return None


def _kernel_name(Q):
    """The scan kernel a batch of Q queries runs (api.hip launch_scan): <= 4 queries per pass stream through the VALU
    kernel, 8..32 through the f32 matrix-core kernel (scan_mfma.hip); larger batches are cut into 32-query passes."""
    qt = _pow2(min(Q, 32))
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
