#!/usr/bin/env python3
"""bench.py — QPS of brute-force top-10 search on the MI355X scorer (BASELINE.json metric), + the other single-GPU configs.

Timed region (the JSON line's `value`).  Workload at N=1 = BASELINE.json configs[1] ("C2"): one segment of 10 M x d=768 f32, cosine,
brute-force exact top-10, resident in HBM.  A *step* = one pass of the hot path over one batch of Q queries: Metric::preprocess of
the batch (qmx_query_update) + one scan of the whole segment with per-query top-k (qmx_search_topk_async =
BatchFilteredSearcher::peek_top_iter) [+ for N>1 the RCCL all-gather of the per-GPU top-k and the k-way merge].  1024 distinct
queries are cycled in batches.

N>1 (torchrun, one rank per GPU):
  --scaling weak (default) = configs[4] ("C5"): rank r holds its own 10 M-row segment (seed + r), every rank scores the same query batch
      against its segment, the per-rank top-k lists (Q x 10 x 8 B) are all-gathered over RCCL/xGMI and merged (BatchResultAggregator
      semantics).  Per-GPU work is fixed.  The counted unit is one (query, 10 M-row segment) search: value = N * Q * steps / time; at
      N=1 this is plain QPS on C2.  The collection-level QPS of the N-segment collection (= value / N) is in config.collection_qps.
  --scaling strong: ONE 10 M-row segment row-split over the ranks, same gather + merge; total work is fixed, value = Q *
      steps / time.

Outside the timed region, rank 0, N=1 (the `configs` object; each entry carries its own roofline, recall@10 against exact search on
the device, and an in-run check of a sample against the CPU oracle):
  C3  10 M x 768 SQ-int8, dot: brute force with oversampling 2 + f32 rescoring at Q = 1 and 32 (qmx_search_quantized), and the HNSW path:
      device build THROUGH the SQ scorer, SQ walk ef = 128, oversampling 2, f32 rescoring.
  TQ4 the rows of C3 as TurboQuant 4-bit: device encode (qmx_tq_encode), brute force with oversampling 2 + f32 rescoring at Q = 1 and 32
      (int8 matrix cores from 4 queries up).
  C4  10 M x 1536 PQ m = 96 (LUT on the matrix cores), HNSW ef = 128: device k-means + encode, device build through the PQ scorer
      (qmx_hnsw_build_quantized), PQ walk, with and without f32 rescoring.
Rows of C3 / C4 have low intrinsic dimension (qmx_synth_fill_latent_f32, 32 latent coordinates + noise): recall is meaningful there.
C2 keeps the iid rows of SURVEY 8d.  Its RESULT is data-independent (the exact top-10, bit for bit); its SPEED is not: the timed path is
a prefilter over a derived copy of the block + exact verification (scan_split.hip: the int8 copy, --split-copy i8, by default; the f16 half copy
is timed beside it, `half_copy_point`), whose verification lists grow where scores crowd near the k-th best (and, for the int8 copy, where
columns carry rare extreme values) and whose overflowing queries take the exact scan.  `robustness` therefore repeats the same search, outside the timed region, on the latent rows of
C3 and on a block with 1 % duplicated rows, and reports candidates / re-scored rows / fallback queries per batch (qmx_counters) next to
QPS and the comparison with the exact scan; `batch_sweep` runs Q in {1, 8, 32, 128} on both tracks (exact f32 stream | prefilter).

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes of one scan launch (the bytes of the block the launched kernel streams: rows x
3072 B for the exact scans, the derived copy's bytes - 768 B per row for the int8 copy, two launches per pass - for the prefilter) / the scan kernel's mean
duration, measured with HIP-event pairs recorded on the kernel's own stream inside the timed region (qmx_query_set_timing /
qmx_query_timing); `roofline.kernel` is the symbol the library reports for the launch (qmx_query_last_kernel).  `cpu_baseline` = the
CPU oracle (AVX2+FMA restatement of the reference's scorer and its peek_top_iter loop) timed on this box's host cores on a bounded
sample of the same rows, single-threaded (the reference's unit of work: one (batch, segment) task) and on every usable core.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak (same guide: v_mfma_f32_16x16x4_f32 / 32x32x2, 64 FLOP/clk/SIMD)
QUERY_ROW0 = 1 << 40   # latent-model queries: rows of the same generator (same basis), far past the stored range


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=128, help="queries per step (Q); more than 64 run 128 per pass through the f16 prefilter + exact verification")
    ap.add_argument("--top", type=int, default=10)
    ap.add_argument("--nqueries", type=int, default=1024)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N>1: weak = one --rows segment per GPU (C5); strong = ONE --rows segment row-split over the GPUs")
    ap.add_argument("--cpu-rows", type=int, default=1_000_000, help="rows of the CPU-baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--split-copy", choices=["auto", "half", "pair", "i8", "none"], default="auto",
                    help="which derived copy of the block the prefilter scans (half: f16 high parts, 2 B / element; pair: f16 pairs, 4 B; i8: int8 codes, 1 B; "
                         "none: the f32 block itself)")
    ap.add_argument("--no-hbm-point", action="store_true", help="skip the secondary Q=16 (HBM-bound) measurement of the same scan")
    ap.add_argument("--no-other-copy-point", action="store_true",
                    help="skip the secondary measurement of the same search over the OTHER derived copy (the f16 half copy when the int8 copy is timed, and vice versa)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the Q in {1, 8, 32, 128} x {exact, prefilter} sweep")
    ap.add_argument("--no-robustness", action="store_true", help="skip the C2 search on latent / duplicated rows (needs 46 GB more HBM)")
    ap.add_argument("--verify", type=int, default=1, help="check samples against the oracle (C2 first batch, C3 / C4 scans and walks)")
    ap.add_argument("--configs", default="c3,tq,c4", help="comma list of the secondary single-GPU configs to run (empty = none)")
    ap.add_argument("--config-rows", type=int, default=0, help="rows of C3 / C4 (0 = --rows)")
    ap.add_argument("--hnsw-queries", type=int, default=8192, help="searches per launch of the HNSW walks")
    ap.add_argument("--in-flight", type=int, default=2, choices=[1, 2, 3, 4],
                    help="query batches in flight on one GPU: 2 = consecutive steps alternate between two query handles on two streams, so that one batch's "
                         "head (preprocess, sample pre-scan, pack) and tail (probe, verification, sort) run beside the other's scans; 1 = one handle, one stream")
    ap.add_argument("--fanout-rows", type=int, default=1_000_000,
                    help="rows per segment of the one-process fan-out legs (qmx_sharded_hnsw_build + qmx_sharded_search_topk over one segment per device); 0 = skip")
    return ap.parse_args()


def usable_cores():
    """Cores this process may actually run on: the affinity mask, cut by the cgroup CPU quota (os.cpu_count() reports the host's)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import numpy as np
    import qdrant_amd as qa
    from qdrant_amd import _ffi as F
    from qdrant_amd import sharded

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        # prove the collective spans N ranks before anything is timed: a real all-gather of one word per rank over RCCL
        probe = torch.zeros(world, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(probe, torch.full((1,), rank + 1, dtype=torch.int32, device=dev))
        torch.cuda.synchronize(dev)
        rccl_ranks = int((probe > 0).sum().item())
        assert probe.tolist() == list(range(1, world + 1)), probe.tolist()

    lib = F.lib()
    dim, Q, top = args.dim, args.batch, args.top
    seed = 0x5EED0002  # 0x5EED0000 + config id
    strong = world > 1 and args.scaling == "strong"
    if strong:
        row0, n = sharded.row_split(args.rows)          # this rank's slice of the ONE segment
        row_seed = seed
    else:
        row0, n = 0, args.rows
        row_seed = seed + 16 * rank                     # this rank's own segment

    # ---- the stored block: generated and normalised on device, adopted without copying ----
    rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
    F.check(lib.qmx_synth_fill_f32(local_rank, row_seed, row0, n, dim, F.ptr(rows)))
    F.check(lib.qmx_preprocess_f32(local_rank, int(qa.Distance.Cosine), F.ptr(rows), n, dim, F.ptr(rows)))
    # (+ the f16-pair copy of the block when batches of more than 64 queries will scan it: scan_split.hip; 4 more bytes per element)
    copy_flag = {"none": 0, "pair": F.SEG_SPLIT_COPY, "half": F.SEG_HALF_COPY, "i8": F.SEG_I8_COPY, "auto": F.SEG_AUTO_COPY}[args.split_copy]
    storage = qa.VectorStorage(rows, qa.Distance.Cosine, device_id=local_rank, flags=copy_flag)
    seg_info = storage.info()       # which copy the segment holds (under "auto": the library's own choice, measured at create) and what the trial saw
    eff_flag = {"i8": F.SEG_I8_COPY, "half": F.SEG_HALF_COPY, "pair": F.SEG_SPLIT_COPY, None: 0}[seg_info["derived_copy"]]

    nbatches = max(1, args.nqueries // Q)
    queries = torch.empty((nbatches * Q, dim), dtype=torch.float32, device=dev)
    F.check(lib.qmx_synth_fill_f32(local_rank, seed + 1, 0, nbatches * Q, dim, F.ptr(queries)))

    stream = torch.cuda.Stream(dev)  # every kernel, the RCCL gather and the merge are ordered on this stream
    torch.cuda.set_stream(stream)
    backend = sharded.HipBackend(storage, Q, local_rank, stream)      # owns the qmx_query of this rank
    qh = backend.qh
    F.check(lib.qmx_query_set_timing(qh, 1))
    searcher = sharded.ShardedSearcher(backend, n, Q, top, device=dev)  # scan -> all-gather -> merge (world > 1)
    out, counts = searcher.out, searcher.counts
    # a second batch in flight (single GPU): its own query handle, stream and result buffers; steps alternate between the two.  Every step is still one
    # whole search of one batch - the GPU merely has the next batch's head to run while this batch's scans and tail are in flight
    lanes = [(backend, out, counts)]
    for _ in range(args.in_flight - 1 if world == 1 else 0):
        backend2 = sharded.HipBackend(storage, Q, local_rank, torch.cuda.Stream(dev))
        F.check(lib.qmx_query_set_timing(backend2.qh, 1))
        lanes.append((backend2, torch.zeros_like(out), torch.zeros_like(counts)))

    def step(i):
        b = i % nbatches
        qb = queries[b * Q:(b + 1) * Q]
        if world > 1:
            searcher.search(qb)
        else:
            be, o, c = lanes[i % len(lanes)]
            with torch.cuda.stream(be.stream):
                be.local_topk(qb, top, o, c)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    fence()
    ms0, l0 = C.c_float(), C.c_uint32()
    for be, _, _ in lanes:
        F.check(lib.qmx_query_timing(be.qh, C.byref(ms0), C.byref(l0)))  # drop warm-up launches

    fence()
    # (the spread of the timed region, without touching it: an event on the stream every `gsz` steps, read after the closing fence)
    gsz = max(1, args.steps // 10)
    marks = [torch.cuda.Event(enable_timing=True)]
    t0 = time.perf_counter()
    marks[0].record(lanes[0][0].stream if world == 1 else stream)
    for i in range(args.steps):
        step(i)
        if (i + 1) % gsz == 0:
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record(lanes[i % len(lanes)][0].stream if world == 1 else stream)
    fence()
    elapsed = time.perf_counter() - t0
    group_ms = [marks[j].elapsed_time(marks[j + 1]) / gsz for j in range(len(marks) - 1)]      # device time per step, per group of gsz steps

    kms, kl = C.c_float(), C.c_uint32()
    for be, _, _ in lanes:       # the scan launches of every batch in flight
        m1, l1 = C.c_float(), C.c_uint32()
        F.check(lib.qmx_query_timing(be.qh, C.byref(m1), C.byref(l1)))
        kms.value += m1.value
        kl.value += l1.value
    kernel_symbol = F.last_kernel(qh)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        km = torch.tensor([kms.value / max(1, kl.value)], dtype=torch.float64, device=dev)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kernel_ms = float(km.item())
    else:
        kernel_ms = kms.value / max(1, kl.value)

    # bytes the dominant kernel has to read per launch: the f32 block (3072 B/row at d=768) for the exact scans; the derived copy
    # the prefilter scans (QMX_SEG_I8_COPY: 1 B / element, QMX_SEG_HALF_COPY: 2 B, QMX_SEG_SPLIT_COPY: 4 B) when that is the kernel that ran
    half_copy = "scan_f16pair_kernel<true>" in kernel_symbol or "scan_f16half256_kernel" in kernel_symbol
    tile_q = 256.0 if "scan_f16half256_kernel" in kernel_symbol else 128.0            # queries per pass of the prefilter shape that ran
    i8_copy = "scan_i8copy_kernel" in kernel_symbol
    elem_bytes = 1 if i8_copy else 2 if half_copy else 4
    row_bytes = dim * elem_bytes
    launches_per_step = max(1, int(kl.value)) / float(max(1, args.steps))
    # the prefilter over a derived copy covers the block in TWO launches of the same kernel (the strided sixteenth of the tiles, then the rest
    # under the threshold the first one tightened): bytes and flops per launch are the per-launch AVERAGES, like kernel_ms, so that
    # achieved = sum of bytes / sum of kernel time
    prefilter = "scan_f16pair_kernel" in kernel_symbol or "scan_f16half256_kernel" in kernel_symbol or i8_copy
    launches_per_pass = max(1.0, launches_per_step / math.ceil(Q / tile_q)) if prefilter else 1.0
    alg_bytes = int(n * row_bytes / launches_per_pass)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    units = Q * args.steps * (1 if strong else world)
    value = units / elapsed

    if world == 1:
        workload = "C2: 1 segment %s x d=%d f32 cosine, brute-force exact top-%d, batch Q=%d" % (_human(n), dim, top, Q)
    elif strong:
        workload = ("C2 row-split: ONE segment %s x d=%d f32 cosine split by contiguous row range over %d GPUs (%s rows each), top-%d, batch Q=%d, "
                    "RCCL all-gather + merge" % (_human(args.rows), dim, world, _human(n), top, Q))
    else:
        workload = ("C5: %d segments (one per GPU) x %s x d=%d f32 cosine, top-%d, batch Q=%d, RCCL all-gather + merge"
                    % (world, _human(n), dim, top, Q))
    result = {
        "metric": "QPS @ recall@10, brute-force, d=%d %s vecs, f32 cosine top-%d" % (dim, _human(args.rows), top),
        "value": round(value, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        # spread over groups of steps inside the timed region (device time between stream events): standard deviation of the groups' QPS
        "value_stddev": round(_stddev([Q * (1 if strong else world) / (m * 1e-3) for m in group_ms if m > 0]), 2),
        "step_groups": {"steps_per_group": gsz, "ms_per_step_min": round(min(group_ms), 4) if group_ms else None,
                        "ms_per_step_max": round(max(group_ms), 4) if group_ms else None},
        "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "rccl_ranks": rccl_ranks,
        "config": {"workload": workload,
                   "rows_per_gpu": n, "dim": dim, "batch": Q, "top": top, "distinct_queries": nbatches * Q,
                   "unit_of_value": ("queries per second against the ONE row-split segment" if strong else
                                     "(query, 10M-row segment) searches per second; at n_gpus=1 this is plain QPS"),
                   "collection_qps": round(Q * args.steps / elapsed, 2),
                   "timed_path": _timed_path(kernel_symbol),
                   "batches_in_flight": len(lanes),
                   "derived_copy": dict(seg_info, requested=args.split_copy)},
        "roofline": _roofline(n, dim, kernel_ms, alg_bytes, achieved, int(kl.value), launches_per_step, Q, kernel_symbol, launches_per_pass),
    }

    solo = rank == 0 and world == 1
    if solo and args.verify and (Q > 64 or copy_flag):
        # the timed path (prefilter + exact verification) against the exact chain-major scan, whole block, first batch: ids and score bits
        step(0)
        torch.cuda.synchronize(dev)
        a_out, a_cnt = out.clone(), counts.clone()
        qa.set_option("no_split_scan", 1)
        try:
            step(0)
            torch.cuda.synchronize(dev)
        finally:
            qa.set_option("no_split_scan", -1)
        result["prefilter_equals_exact_scan_whole_block"] = bool(torch.equal(a_out, out) and torch.equal(a_cnt, counts))
        # recall@10 of the timed path against the exact scan (ids as sets per query): 1.0 by construction, measured anyway
        hit = sum(len(set(a_out[i, :, 0].tolist()) & set(out[i, :, 0].tolist())) for i in range(Q))
        result["recall_at_10"] = round(hit / float(Q * top), 6)
    if solo and Q != 16 and not args.no_hbm_point:
        # the HBM-bound operating point of the same scan (north_star: >= 70 % of the HBM roofline on C2): 16 queries per pass, where the
        # kernel is a pure stream of the stored block; outside the timed region, same rows, same measurement (HIP events on the kernel's stream)
        qa.set_option("no_split_scan", 1)      # (this point is the EXACT scan's: the f32 block itself, streamed once for 16 queries)
        try:
            result["roofline_hbm_point_q16"] = hbm_point(16, storage, queries, n, dim, top, local_rank, stream, lib, F, sharded, torch)
        except Exception as e:
            result["roofline_hbm_point_q16"] = {"error": repr(e)[:300]}
        finally:
            qa.set_option("no_split_scan", -1)
    if solo and Q != 256 and eff_flag == F.SEG_HALF_COPY and queries.shape[0] >= 256 and not args.no_hbm_point:
        # the same search at 256 queries per step: the 256-query shape of the prefilter (half the copy's bytes per query; issue-bound, not
        # HBM-bound: more queries per second at a lower fraction of either roof)
        try:
            p = hbm_point(256, storage, queries, n, dim, top, local_rank, stream, lib, F, sharded, torch, bytes_per_pass=n * dim * 2)
            tfl = 2.0 * n * dim * 256 / p["launches_per_pass"] / (p["kernel_ms"] * 1e-3) / 1e12 if p["kernel_ms"] > 0 else 0.0
            p["mfma_f16"] = {"achieved_TFLOPs": round(tfl, 1), "peak_TFLOPs": MFMA_F16_PEAK_TFLOPS, "frac": round(tfl / MFMA_F16_PEAK_TFLOPS, 4)}
            result["throughput_point_q256"] = p
        except Exception as e:
            result["throughput_point_q256"] = {"error": repr(e)[:300]}
    if solo and eff_flag in (F.SEG_I8_COPY, F.SEG_HALF_COPY) and not args.no_other_copy_point and queries.shape[0] >= Q:
        # the same search over the OTHER derived copy of the block - the f16 half copy (2 B / element, band 1e-3 |q| |row|: the round-2 / round-3
        # headline) when the int8 copy (1 B / element, worst-case band of the two roundings) is the timed one, and vice versa: lists checked against
        # the exact scan inside the leg; a secondary point
        other = F.SEG_HALF_COPY if eff_flag == F.SEG_I8_COPY else F.SEG_I8_COPY
        key = "half_copy_point" if other == F.SEG_HALF_COPY else "int8_copy_point"
        try:
            result[key] = derived_copy_point(other, rows, queries, n, dim, Q, top, local_rank, stream, lib, F, qa, sharded, torch)
        except Exception as e:
            result[key] = {"error": repr(e)[:300]}
    if solo and not args.no_hbm_point and not args.no_sweep:
        # BASELINE.md / SURVEY 8d name Q in {1, 8, 32}: both tracks at each batch size, same rows, same measurement
        sweep = {}
        for Qs in (1, 8, 32, 128):
            if queries.shape[0] < Qs:
                continue
            for track in (("exact", "prefilter") if eff_flag else ("exact",)):
                qa.set_option("no_split_scan", 1 if track == "exact" else -1)
                try:
                    bpp = None if track == "exact" else n * dim * (1 if eff_flag == F.SEG_I8_COPY else 2 if eff_flag == F.SEG_HALF_COPY else 4)
                    sweep["Q%d_%s" % (Qs, track)] = hbm_point(Qs, storage, queries, n, dim, top, local_rank, stream, lib, F, sharded, torch, bytes_per_pass=bpp, steps=20)
                except Exception as e:
                    sweep["Q%d_%s" % (Qs, track)] = {"error": repr(e)[:300]}
                finally:
                    qa.set_option("no_split_scan", -1)
        result["batch_sweep"] = sweep
    if solo and copy_flag and not args.no_robustness:
        try:
            result["robustness"] = robustness(args, dev, rows, queries, n, dim, Q, top, local_rank, stream, copy_flag, lib, F, qa, sharded, torch)
        except Exception as e:
            result["robustness"] = {"error": repr(e)[:400]}
    if solo:
        c = F.Counters()
        try:   # the prefilter's own counters on the timed rows (one more batch, outside the timed region)
            step(0)
            F.check(lib.qmx_query_last_counters(qh, C.byref(c)))
            result["roofline"]["prefilter_per_batch"] = _counters_dict(c, Q)
        except Exception as e:
            result["roofline"]["prefilter_per_batch"] = {"error": repr(e)[:200]}
    if rank == 0 and args.fanout_rows > 0:
        # the ONE-PROCESS fan-out behind the C-ABI (what a Rust host that owns all segments of a node calls): index build over independent segments
        # and the sharded search with its merge, over one segment per visible device of this run (world > 1: the other ranks wait at the barrier
        # below; a single GPU: two segments on it, which exercises the same code path)
        try:
            result["one_process_fanout"] = one_process_fanout(args, world, dim, Q, top, lib, F, qa, torch, np)
        except Exception as e:
            result["one_process_fanout"] = {"error": repr(e)[:400]}
    if world > 1:
        dist.barrier()
    if solo and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(args, rows, queries, out, counts, n, dim, Q, top, lib, qh, F, qa, np, torch)
    for be, _, _ in lanes[1:]:
        be.close()
    backend.close()
    wanted = [c for c in args.configs.lower().split(",") if c]
    if solo and wanted:
        cfg = {}
        del searcher, backend, storage, out, counts
        ctx = dict(args=args, dev=dev, lib=lib, F=F, qa=qa, np=np, torch=torch)
        if "c3" in wanted:
            try:
                cfg["C3"], rows = c3_section(ctx, rows)
            except Exception as e:  # the headline line must survive a failure of a secondary measurement
                cfg["C3"] = {"error": repr(e)[:400]}
        if "tq" in wanted and "C3" in cfg and "error" not in cfg["C3"]:
            try:
                cfg["TQ4"] = tq_section(ctx, rows)      # the rows of C3, TurboQuant 4 bits instead of SQ int8
            except Exception as e:
                cfg["TQ4"] = {"error": repr(e)[:400]}
        del rows
        torch.cuda.empty_cache()
        if "c4" in wanted:
            try:
                cfg["C4"] = c4_section(ctx)
            except Exception as e:
                cfg["C4"] = {"error": repr(e)[:400]}
        result["configs"] = cfg
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------------
# C2 helpers
# ------------------------------------------------------------------------------------------------------------------------
def _counters_dict(c, Q):
    """qmx_counters of one batch -> what the prefilter cost on this data (zeros on the exact track)."""
    pq = max(1, int(c.prefilter_queries))
    return {"prefilter_queries": int(c.prefilter_queries), "candidates_per_query": round(c.prefilter_candidates / float(pq), 1),
            "verified_rows_per_query": round(c.verified_rows / float(pq), 1), "fallback_queries": int(c.fallback_queries),
            "fallback_rate": round(c.fallback_queries / float(pq), 4), "bytes_read": int(c.bytes_read)}


def _family_rows(torch, dev, kind, n, dim, seed, out=None, chunk=1_000_000):
    """Unit rows of the families of DESIGN 3.1e on which the int8 copy's worst-case band is widest: 'student5' (heavy-tailed elements: Student t, 5 degrees
    of freedom) and 'dominant8' (Gaussian with 8 coordinates twelve times the others).  Generated on the device in chunks (torch's generator: harness only)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    buf = out if out is not None else torch.empty((n, dim), dtype=torch.float32, device=dev)
    for r0 in range(0, n, chunk):
        m = min(chunk, n - r0)
        x = torch.randn((m, dim), generator=g, device=dev, dtype=torch.float32)
        if kind == "student5":
            chi = torch.zeros((m, dim), device=dev, dtype=torch.float32)
            for _ in range(5):
                chi += torch.randn((m, dim), generator=g, device=dev, dtype=torch.float32) ** 2
            x = x / torch.sqrt(chi / 5.0)
            del chi
        elif kind == "dominant8":
            x[:, :8] *= 12.0
        buf[r0:r0 + m] = x / x.norm(dim=1, keepdim=True)
    return buf


def _robust_leg(buf, qs, flag, Q, top, local_rank, stream, lib, F, qa, sharded, torch, dev, what, check_all=True, deleted=None):
    """One timed search of `qs` (batches of Q) over `buf` with the derived-copy flag `flag`: QPS, the prefilter's counters per batch, which copy the
    segment holds (qmx_segment_get_info) and whether every list equals the exact scan's, bit for bit."""
    st = qa.VectorStorage(buf, qa.Distance.Cosine, device_id=local_rank, flags=flag)
    if deleted is not None:
        st.set_deleted(deleted)
    backend = sharded.HipBackend(st, Q, local_rank, stream)
    try:
        o = torch.zeros((Q, top, 2), dtype=torch.int32, device=dev)
        cn = torch.zeros((Q,), dtype=torch.int32, device=dev)
        nb = max(1, qs.shape[0] // Q)
        for i in range(3):
            backend.local_topk(qs[(i % nb) * Q:(i % nb + 1) * Q], top, o, cn)
        torch.cuda.synchronize(dev)
        steps = 20
        t0 = time.perf_counter()
        for i in range(steps):
            backend.local_topk(qs[(i % nb) * Q:(i % nb + 1) * Q], top, o, cn)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        per_batch, same = [], True
        kernel = F.last_kernel(backend.qh)
        for b in range(nb if check_all else 1):
            backend.local_topk(qs[b * Q:(b + 1) * Q], top, o, cn)
            c = F.Counters()
            F.check(lib.qmx_query_last_counters(backend.qh, C.byref(c)))
            per_batch.append(_counters_dict(c, Q))
            a_o, a_c = o.clone(), cn.clone()
            qa.set_option("no_split_scan", 1)
            try:
                backend.local_topk(qs[b * Q:(b + 1) * Q], top, o, cn)
                torch.cuda.synchronize(dev)
            finally:
                qa.set_option("no_split_scan", -1)
            same = same and bool(torch.equal(a_o, o) and torch.equal(a_c, cn))
        nchk = len(per_batch)
        info = st.info()
        return {"rows": what, "batch": Q, "qps": round(Q * steps / wall, 1), "ms_per_step": round(wall / steps * 1e3, 4), "kernel": kernel,
                "copy": info["derived_copy"], "copy_chosen_by_trial": info["chosen_by_trial"], "i8_scale_balance": round(info["i8_scale_balance"], 2),
                "trial": ({"i8_ms": round(info["trial_i8_ms"], 3), "half_ms": round(info["trial_half_ms"], 3),
                           "i8_verified_rows_per_query": round(info["trial_i8_verified_rows"], 1),
                           "i8_fallback_queries": info["trial_i8_fallback_queries"]} if info["chosen_by_trial"] else None),
                "batches_checked": nchk, "equals_exact_scan_whole_block": same,
                "candidates_per_query": round(sum(p["candidates_per_query"] for p in per_batch) / nchk, 1),
                "verified_rows_per_query": round(sum(p["verified_rows_per_query"] for p in per_batch) / nchk, 1),
                "fallback_queries_per_batch": [p["fallback_queries"] for p in per_batch],
                "fallback_rate": round(sum(p["fallback_queries"] for p in per_batch) / float(nchk * Q), 4)}
    finally:
        backend.close()
        st.close()


def robustness(args, dev, c2_rows, queries_iid, n, dim, Q, top, local_rank, stream, copy_flag, lib, F, qa, sharded, torch):
    """The timed search (same Q) on rows that are not the friendly iid block:
      (a) the latent rows of C3 (32 latent coordinates + noise, queries from the same model) and (b) the iid block with 1 % of its rows overwritten by
          copies of 1 000 source rows (100 copies each) - rows where scores crowd -, through the timed copy flag;
      (c) SURVEY 8(d)'s run with 1 % random deleted bits on the C2 block;
      (d) the families on which the int8 copy's worst-case band is widest - Student-t(5) elements, 8 dominant coordinates - through the
          int8 copy, the half copy and QMX_SEG_AUTO_COPY (the library's own choice, measured at create): QPS, fallback rate, verified rows per family
          and copy, and which copy AUTO kept.
    Every leg reports whether every list equals the exact scan's, bit for bit."""
    out = {}
    buf = torch.empty((n, dim), dtype=torch.float32, device=dev)
    seed = 0x5EED0003
    leg = lambda b, qs, flag, what, **kw: _robust_leg(b, qs, flag, Q, top, local_rank, stream, lib, F, qa, sharded, torch, dev, what, **kw)
    nqs = max(Q, 256)
    # (a) latent rows + latent queries
    F.check(lib.qmx_synth_fill_latent_f32(local_rank, seed, 0, n, dim, 32, 1.0, F.ptr(buf)))
    F.check(lib.qmx_preprocess_f32(local_rank, int(qa.Distance.Cosine), F.ptr(buf), n, dim, F.ptr(buf)))
    ql = torch.empty((nqs, dim), dtype=torch.float32, device=dev)
    F.check(lib.qmx_synth_fill_latent_f32(local_rank, seed, QUERY_ROW0, ql.shape[0], dim, 32, 1.0, F.ptr(ql)))
    torch.cuda.synchronize(dev)
    out["latent_rows_of_C3"] = leg(buf, ql, copy_flag, "10M x 768 rows of low intrinsic dimension (32 latent coordinates + noise), queries of the same model")
    # (b) duplicates
    g = torch.Generator(device="cpu").manual_seed(1234)
    n_dup = n // 100
    src = torch.randint(0, n, (1000,), generator=g)
    dst = torch.randperm(n, generator=g)[:n_dup]
    buf.copy_(c2_rows)
    buf[dst.to(dev)] = c2_rows[src.to(dev)].repeat_interleave(n_dup // 1000, dim=0)[:n_dup]
    qs = queries_iid[:nqs].clone()
    half = qs.shape[0] // 2
    qs[:half] = c2_rows[src[:half].to(dev)] + 0.02 * qs[:half]
    torch.cuda.synchronize(dev)
    out["iid_with_1pct_duplicates"] = leg(buf, qs, copy_flag, "the C2 block with 1 % of its rows overwritten by copies of 1000 source rows (100 copies each); "
                                          "half of the queries are noisy copies of source rows, so their best scores are 100-fold ties")
    # (c) SURVEY 8(d): 1 % random deleted bits on the C2 block itself
    import numpy as np
    deleted = np.random.default_rng(77).random(n) < 0.01
    out["c2_with_1pct_deleted"] = leg(c2_rows, queries_iid[:nqs], copy_flag, "the C2 block with 1 % of its points deleted at random (SURVEY 8d)", deleted=deleted)
    out["c2_with_1pct_deleted"]["deleted_points"] = int(deleted.sum())
    # (d) the hard families, every copy + the library's own choice
    flags = (("int8_copy", F.SEG_I8_COPY), ("half_copy", F.SEG_HALF_COPY), ("auto_copy", F.SEG_AUTO_COPY))
    for kind, what in (("student5", "10M x 768 unit rows with Student-t(5) elements (heavy tails: column maximum / column spread ~ 25)"),
                       ("dominant8", "10M x 768 unit Gaussian rows with 8 coordinates twelve times the others (60 % of the score lives on 8 columns)")):
        _family_rows(torch, dev, kind, n, dim, 0xFA0000 + len(kind), out=buf)
        qf = _family_rows(torch, dev, kind, nqs, dim, 0xFA1000 + len(kind))
        torch.cuda.synchronize(dev)
        fam = {}
        for name, flag in flags:
            try:
                fam[name] = leg(buf, qf, flag, what, check_all=False)
            except Exception as e:
                fam[name] = {"error": repr(e)[:300]}
        ok = [k for k in fam if "qps" in fam[k]]
        if "auto_copy" in ok and len(ok) == 3:
            fam["auto_vs_best_fixed"] = round(fam["auto_copy"]["qps"] / max(fam["int8_copy"]["qps"], fam["half_copy"]["qps"]), 3)
        out[kind] = fam
    del buf
    torch.cuda.empty_cache()
    return out


def one_process_fanout(args, world, dim, Q, top, lib, F, qa, torch, np):
    """north_star's multi-GPU sentence behind the C-ABI, from ONE host process: `qmx_sharded_hnsw_build` (one host thread per segment inside the
    library, each on its segment's device: the reference locks one GPU of its pool per segment build, gpu_devices_manager.rs:120-143) and
    `qmx_sharded_search_topk` (per-device scans enqueued side by side, per-segment lists copied to the first device over xGMI, merged there:
    segments_searcher.rs:250-285 + search_result_aggregator.rs:50-121).  Segments: `--fanout-rows` x dim f32 cosine each, one per device (two on the
    only device of a 1-GPU run).  Reports points/s of the build fan-out against the same builds one after the other, and QPS of the sharded search
    against one segment alone; the merged lists are checked against the per-segment searches merged on the host."""
    n = args.fanout_rows
    devs = list(range(world)) if world > 1 else [0, 0]
    nseg = len(devs)
    rows, storages = [], []
    for i, d in enumerate(devs):
        r = torch.empty((n, dim), dtype=torch.float32, device=torch.device("cuda", d))
        F.check(lib.qmx_synth_fill_f32(d, 0x5EED0500 + i, 0, n, dim, F.ptr(r)))
        F.check(lib.qmx_preprocess_f32(d, int(qa.Distance.Cosine), F.ptr(r), n, dim, F.ptr(r)))
        rows.append(r)
        storages.append(qa.VectorStorage(r, qa.Distance.Cosine, device_id=d, flags=F.SEG_AUTO_COPY))
    for d in set(devs):
        torch.cuda.synchronize(d)
    out = {"segments": nseg, "devices": sorted(set(devs)), "rows_per_segment": n, "dim": dim}
    # ---- build fan-out ----
    kw = dict(m=16, ef_construct=100, seed=42)
    t0 = time.perf_counter()
    graphs = qa.GraphLayers.build_sharded(storages, **kw)
    t_fan = time.perf_counter() - t0
    t0 = time.perf_counter()
    g_one = qa.GraphLayers.build(storages[0], **kw)
    t_one = time.perf_counter() - t0
    g_one.close()
    out["build"] = {"what": "qmx_sharded_hnsw_build: HNSW m=16 ef_construct=100 over every segment at once, one host thread per segment",
                    "seconds": round(t_fan, 3), "points_per_s": round(nseg * n / t_fan, 1),
                    "one_segment_alone_seconds": round(t_one, 3), "one_segment_alone_points_per_s": round(n / t_one, 1),
                    "speedup_over_sequential": round(nseg * t_one / t_fan, 3)}
    # ---- sharded search ----
    qs = torch.empty((Q, dim), dtype=torch.float32, device=torch.device("cuda", devs[0]))
    F.check(lib.qmx_synth_fill_f32(devs[0], 0x5EED0501, 0, Q, dim, F.ptr(qs)))
    torch.cuda.synchronize(devs[0])
    qh_host = qs.cpu().numpy()
    handles = []
    for st in storages:
        h = C.c_void_p()
        F.check(lib.qmx_query_create(st._h, F.ptr(qh_host), Q, C.byref(h)))
        handles.append(h)
    arr = (C.c_void_p * nseg)(*[h.value for h in handles])
    bases = np.arange(nseg, dtype=np.uint32) * np.uint32(n)
    merged = np.zeros((Q, top), dtype=np.dtype([("idx", np.uint32), ("score", np.float32)]))
    mcnt = np.zeros(Q, dtype=np.uint32)
    steps = 20
    for _ in range(3):
        F.check(lib.qmx_sharded_search_topk(arr, nseg, top, F.ptr(bases), F.ptr(merged), F.ptr(mcnt), None, None))
    t0 = time.perf_counter()
    for _ in range(steps):
        F.check(lib.qmx_sharded_search_topk(arr, nseg, top, F.ptr(bases), F.ptr(merged), F.ptr(mcnt), None, None))
    t_sh = (time.perf_counter() - t0) / steps
    one = np.zeros_like(merged)
    ocnt = np.zeros(Q, dtype=np.uint32)
    for _ in range(3):
        F.check(lib.qmx_search_topk(handles[0], top, None, 0, F.ptr(one), F.ptr(ocnt), None, None))
    t0 = time.perf_counter()
    for _ in range(steps):
        F.check(lib.qmx_search_topk(handles[0], top, None, 0, F.ptr(one), F.ptr(ocnt), None, None))
    t_1 = (time.perf_counter() - t0) / steps
    # the merged lists against the per-segment lists merged on the host (descending score, lower global id first among equals)
    per = []
    for i, h in enumerate(handles):
        o = np.zeros_like(merged)
        c = np.zeros(Q, dtype=np.uint32)
        F.check(lib.qmx_search_topk(h, top, None, 0, F.ptr(o), F.ptr(c), None, None))
        o["idx"] += np.uint32(i * n)
        per.append(o)
    allp = np.concatenate(per, axis=1)
    same = True
    for qi in range(Q):
        order = np.lexsort((allp[qi]["idx"], -allp[qi]["score"].astype(np.float64)))[:top]
        same = same and np.array_equal(allp[qi][order], merged[qi])
    out["search"] = {"what": "qmx_sharded_search_topk: %d segments x %s rows, batch Q=%d, top-%d, host-synchronous (lists back on the host)" % (nseg, _human(n), Q, top),
                     "ms_per_batch": round(t_sh * 1e3, 4), "qps_collection": round(Q / t_sh, 1), "segment_searches_per_s": round(nseg * Q / t_sh, 1),
                     "one_segment_alone_ms": round(t_1 * 1e3, 4), "efficiency_vs_one_segment": round(t_1 / t_sh if world > 1 else nseg * t_1 / t_sh, 3),
                     "merged_equals_host_merge": bool(same)}
    for h in handles:
        lib.qmx_query_destroy(h)
    for g in graphs:
        g.close()
    for st in storages:
        st.close()
    del rows
    return out


def hbm_point(Qh, storage, queries, n, dim, top, local_rank, stream, lib, F, sharded, torch, bytes_per_pass=None, steps=30):
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
        t0 = time.perf_counter()
        for i in range(steps):
            backend.local_topk(queries[(i % nb) * Qh:(i % nb + 1) * Qh], top, out, counts)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        F.check(lib.qmx_query_timing(backend.qh, C.byref(ms), C.byref(nl)))
        kernel_ms = ms.value / max(1, nl.value)
        per_step = max(1.0, nl.value / float(steps))            # launches per pass over the block (the prefilter over a derived copy: 2)
        # exact track: every launch streams the whole f32 block (a batch of more than 64 queries is several such passes); prefilter: the derived copy
        # is covered by the pass's two launches, the figure is their mean
        alg = int(bytes_per_pass / per_step) if bytes_per_pass else n * dim * 4
        gbps = alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        sym = F.last_kernel(backend.qh)
        return _attach_traffic({"batch": Qh, "kernel": sym, "kernel_ms": round(kernel_ms, 4), "launches_timed": int(nl.value), "launches_per_pass": per_step,
                                "algorithmic_bytes_per_launch": alg, "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": None, "qps": round(Qh * steps / wall, 1),
                                "ms_per_step": round(wall / steps * 1e3, 4)}, sym, n)
    finally:
        backend.close()


def derived_copy_point(flag, rows, queries, n, dim, Q, top, local_rank, stream, lib, F, qa, sharded, torch):
    """The timed search of the headline (same rows, same Q) through another derived copy of the block (QMX_SEG_I8_COPY / QMX_SEG_HALF_COPY): QPS
    (wall), the scan kernel against the HBM roof on the bytes of THAT copy (HIP events on the kernel's stream), what the prefilter let through, and
    whether every list of the first batch equals the exact scan's, bit for bit."""
    i8 = flag == F.SEG_I8_COPY
    st = qa.VectorStorage(rows, qa.Distance.Cosine, device_id=local_rank, flags=flag)      # (adopts the device block; + 1 or 2 B / element)
    try:
        p = hbm_point(Q, st, queries, n, dim, top, local_rank, stream, lib, F, sharded, torch, bytes_per_pass=n * dim * (1 if i8 else 2), steps=50)
        if p["kernel_ms"] > 0:
            tops = 2.0 * n * dim * 128 / p["launches_per_pass"] / (p["kernel_ms"] * 1e-3) / 1e12
            peak = MFMA_I8_PEAK_TOPS if i8 else MFMA_F16_PEAK_TFLOPS
            p["mfma_i8" if i8 else "mfma_f16"] = {"achieved_TOPs": round(tops, 1), "peak_TOPs": peak, "frac": round(tops / peak, 4)}
        backend = sharded.HipBackend(st, Q, local_rank, stream)
        try:
            o = torch.zeros((Q, top, 2), dtype=torch.int32, device=queries.device)
            cn = torch.zeros((Q,), dtype=torch.int32, device=queries.device)
            backend.local_topk(queries[:Q], top, o, cn)
            torch.cuda.synchronize()
            c = F.Counters()
            F.check(lib.qmx_query_last_counters(backend.qh, C.byref(c)))
            p["prefilter_per_batch"] = _counters_dict(c, Q)
            a_o, a_c = o.clone(), cn.clone()
            qa.set_option("no_split_scan", 1)
            try:
                backend.local_topk(queries[:Q], top, o, cn)
                torch.cuda.synchronize()
            finally:
                qa.set_option("no_split_scan", -1)
            p["equals_exact_scan_whole_block"] = bool(torch.equal(a_o, o) and torch.equal(a_c, cn))
        finally:
            backend.close()
        p["copy"] = ("int8 codes, one scale per column and per query, 1 B / element (+ 25 % of the block in HBM); worst-case band, exact bounds renewed after each launch"
                     if i8 else "f16 high parts, 2 B / element (+ 50 % of the block in HBM); band 1e-3 |q| |row|")
        return p
    finally:
        st.close()


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
    cores = usable_cores()

    def run(threads, budget):
        reps, t0 = 0, time.perf_counter()
        while True:
            res = ost.peek_top(enc, top, encoded=True, threads=threads)
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget or reps >= 1000:
                return res, reps, el
    # (a) the reference's unit of work: one thread runs one (query batch, segment) task; a smaller sample keeps it bounded
    S1 = min(S, 100_000)
    ost1 = O.DenseStorage(O.F32, O.COSINE, host_rows[:S1])
    reps1, t0 = 0, time.perf_counter()
    while True:
        ost1.peek_top(enc, top, encoded=True, threads=0)
        reps1 += 1
        el1 = time.perf_counter() - t0
        if el1 >= args.cpu_seconds * 0.4 or reps1 >= 1000:
            break
    qps1 = Q * reps1 / el1 * (S1 / n)
    # (b) every usable core on disjoint row ranges (the reference's segment-parallel model)
    res, reps, el = run(cores, args.cpu_seconds * 0.6)
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
                 np.array_equal(gs[i].view(np.uint32), res[i]["score"].view(np.uint32)) for i in range(Q))
        if not ok:
            print("PARITY FAILURE: GPU top-k differs from the oracle on the CPU sample", file=sys.stderr)
    flops = 2.0 * dim
    return {"value": round(cpu_qps, 3), "unit": "queries/s", "cores": cores, "kind": "port",
            "kind_note": "the oracle's C restatement of the reference's AVX2+FMA scorer and peek_top_iter loop, not the Rust binary (no cargo in the image)",
            "sample": "oracle peek_top_iter (AVX2+FMA dot, 64-id chunks, heap of %d) over the first %d of %d rows, Q=%d, "
                      "%d threads on disjoint row ranges (usable cores: affinity + cgroup quota; os.cpu_count() = %d), %d scans in %.1f s; "
                      "QPS scaled by %d/%d to the full segment" % (top, S, n, Q, cores, os.cpu_count() or 0, reps, el, S, n),
            "gflops_all_cores": round(flops * S * Q * reps / el / 1e9, 1),
            "single_thread": {"value": round(qps1, 4), "unit": "queries/s", "cores": 1,
                              "gflops": round(flops * S1 * Q * reps1 / el1 / 1e9, 2), "ns_per_row_per_query": round(el1 / (reps1 * S1 * Q) * 1e9, 2),
                              "sample": "the same loop, one thread, first %d rows, %d scans in %.1f s, scaled by %d/%d" % (S1, reps1, el1, S1, n)},
            "gpu_matches_oracle_on_sample_bit_exact": ok}


MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16 / bf16 MFMA peak (same guide); 2377 measured in a bare loop (profiles/r2_mfma_issue_rates.txt)
MFMA_I8_PEAK_TOPS = 5000.0  # dense int8 MFMA peak (same guide: twice the f16 rate; v_mfma_i32_16x16x64_i8)


def _roofline(n, dim, kernel_ms, alg_bytes, achieved_gbps, launches, launches_per_step, Q, kernel_symbol, launches_per_pass=1.0):
    """The dominant kernel against BOTH ceilings; `bound` is the one it sits closer to.  Up to 16 queries per pass the scan is
    an HBM stream (every row byte read once: SURVEY 8d, 3072 B / row at d = 768); the 32- / 64-query passes of scan_mfma16.hip
    do 2 * dim flops per (row, query) on the f32 matrix cores and cross over to the MFMA ceiling; the prefilter of scan_split.hip
    (more than 64 queries) streams a derived f16 copy of the block and multiplies on the f16 matrix cores (1 or 3 products per element)."""
    per_pass = Q / max(1.0, round(launches_per_step / launches_per_pass))   # queries one pass over the block serves: MEASURED launches per step, not a dispatch guess
    half256 = "scan_f16half256_kernel" in kernel_symbol
    i8 = "scan_i8copy_kernel" in kernel_symbol
    split = "scan_f16pair_kernel" in kernel_symbol or "scan_f32_split_kernel" in kernel_symbol or half256 or i8
    products = 1 if ("scan_f16pair_kernel<true>" in kernel_symbol or half256 or i8) else 3 if split else 1
    flops = 2.0 * n * dim * ((256 if half256 else 128) if split else per_pass) * products / launches_per_pass     # (the prefilter multiplies a padded 128- / 256-query tile)
    tflops = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    mfma_peak = MFMA_I8_PEAK_TOPS if i8 else MFMA_F16_PEAK_TFLOPS if split else MFMA_F32_PEAK_TFLOPS
    hbm_frac, mfma_frac = achieved_gbps / HBM_PEAK_GBPS, tflops / mfma_peak
    traffic, traffic_src = _pmc_traffic(n, dim, Q, kernel_symbol)
    common = {"traffic": traffic, "traffic_source": traffic_src, "traffic_over_algorithmic": round(traffic / float(alg_bytes), 4) if traffic and alg_bytes else None, "kernel": kernel_symbol, "kernel_ms": round(kernel_ms, 4), "launches_timed": launches,
              "queries_per_pass": per_pass, "launches_per_pass": launches_per_pass, "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_flops_per_launch": flops,
              "hbm": {"achieved_GBps": round(achieved_gbps, 1), "peak_GBps": HBM_PEAK_GBPS, "frac": round(hbm_frac, 4)},
              "mfma": {"dtype": "int8 (column- and query-scaled codes, i32 accumulate; results re-scored exactly in f32)" if i8 else
                                "f16 (x = h + l prefilter, f32 accumulate; results re-scored exactly in f32)" if split else "f32",
                       "achieved_TFLOPs": round(tflops, 2), "peak_TFLOPs": mfma_peak, "frac": round(mfma_frac, 4)},
              "f32_block_equivalent_GBps": round(n * dim * 4 / (kernel_ms * launches_per_pass * 1e-3) / 1e9, 1) if kernel_ms > 0 else 0.0}
    if split:
        eq = common["f32_block_equivalent_GBps"]
        common["frac_of_copy_stream"] = round(hbm_frac, 4)
        common["f32_block_equivalent"] = {"GBps": eq, "frac_of_peak": round(eq / HBM_PEAK_GBPS, 4),
                                          "note": "SURVEY 8(d) counts 4 B / element of the stored f32 block per scan; those bytes are NOT streamed by this kernel: it streams a derived "
                                                  + ("int8" if i8 else "f16") + " copy (achieved / frac above are bytes of the copy / kernel time) and re-scores the survivors from the f32 rows.  The 8(d)-conformant "
                                                  "figure (the f32 block itself streamed once) is roofline_hbm_point_q16 / batch_sweep.Q*_exact."}
    if mfma_frac > hbm_frac:
        return dict({"bound": "mfma", "achieved": round(tflops, 2), "peak": mfma_peak, "unit": "TFLOP/s", "frac": round(mfma_frac, 4)}, **common)
    return dict({"bound": "hbm", "achieved": round(achieved_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(hbm_frac, 4)}, **common)


def _stddev(xs):
    if len(xs) < 2:
        return 0.0
    m = sum(xs) / len(xs)
    return math.sqrt(sum((x - m) ** 2 for x in xs) / (len(xs) - 1))


def _timed_path(kernel_symbol):
    """What the timed step is, in words, from the symbol of the kernel that ran (config.timed_path of the JSON line)."""
    what = None
    if "scan_i8copy_kernel" in kernel_symbol:
        what = "prefilter over an int8 copy of the block (1 B / element, int8 matrix cores)"
    elif "scan_f16pair_kernel<true>" in kernel_symbol or "scan_f16half256_kernel" in kernel_symbol:
        what = "prefilter over an f16 copy of the block (2 B / element, f16 matrix cores)"
    elif "scan_f16pair_kernel" in kernel_symbol:
        what = "prefilter over an f16-pair copy of the block (4 B / element, f16 matrix cores)"
    elif "scan_f32_split_kernel" in kernel_symbol:
        what = "prefilter converting the f32 rows to f16 pairs on the fly (f16 matrix cores)"
    if what is None:
        return "exact f32 scan"
    return what + " + exact f32 re-scoring of the survivors: the returned lists are the exact f32 scan's, bit for bit (checked in the run)"


def _pmc_entry(kernel_symbol):
    """The entry of profiles/pmc_traffic.json for this kernel symbol (template arguments included), or None.  HBM bytes per launch come from
    separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over tools/traffic_workloads.py (counters cannot be read from inside the
    process)."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        for k, e in json.load(open(p)).get("by_kernel", {}).items():
            if _same_kernel(k, kernel_symbol):
                return e
    except Exception:
        pass
    return None


def _pmc_traffic(n, dim, Q, kernel_symbol):
    """(bytes per launch, source) when the table holds THIS kernel symbol at THIS row count; otherwise (None, None): a stale number is worse than none."""
    e = _pmc_entry(kernel_symbol)
    if e and e.get("rows") == n and "over_algorithmic" not in e:
        return e["bytes"], "%s (rocprofv3 --pmc passes on this kernel: %s)" % (e.get("profile", "profiles/pmc_traffic.json"), e.get("workload", ""))
    return None, None


def _attach_traffic(roof, kernel_symbol, n):
    """fills roof['traffic'] (+ the ratio to the algorithmic bytes) from the PMC table; graph walks carry the ratio measured on a smaller graph"""
    e = _pmc_entry(kernel_symbol)
    if not e:
        return roof
    if "over_algorithmic" in e:
        roof["traffic_measured_elsewhere"] = {"rows": e["rows"], "searches": e.get("searches"), "bytes_per_launch": e["bytes"],
                                              "algorithmic_bytes_per_launch": e.get("algorithmic_bytes"), "over_algorithmic": e["over_algorithmic"],
                                              "source": e.get("profile"), "note": e.get("workload")}
    elif e.get("rows") == n:
        roof["traffic"] = e["bytes"]
        roof["traffic_source"] = e.get("profile")
        alg = roof.get("algorithmic_bytes_per_launch")
        if alg:
            roof["traffic_over_algorithmic"] = round(e["bytes"] / float(alg), 4)
    return roof


def _same_kernel(a, b):
    norm = lambda s: "".join(str(s).replace("void ", "").split())   # noqa: E731
    a, b = norm(a), norm(b)
    return a.split("(")[0] == b.split("(")[0]


def _human(n):
    return ("%dM" % (n // 1_000_000)) if n % 1_000_000 == 0 else ("%dk" % (n // 1000)) if n % 1000 == 0 else str(n)


# ------------------------------------------------------------------------------------------------------------------------
# C3 / C4 (rank 0, N = 1, outside the timed region)
# ------------------------------------------------------------------------------------------------------------------------
def _latent(ctx, seed, row0, count, dim, out=None):
    lib, F, qa, torch, dev = ctx["lib"], ctx["F"], ctx["qa"], ctx["torch"], ctx["dev"]
    x = out if out is not None else torch.empty((count, dim), dtype=torch.float32, device=dev)
    F.check(lib.qmx_synth_fill_latent_f32(dev.index or 0, seed, row0, count, dim, 32, 1.0, F.ptr(x)))
    F.check(lib.qmx_preprocess_f32(dev.index or 0, int(qa.Distance.Cosine), F.ptr(x), count, dim, F.ptr(x)))
    return x


def _recall(got, exact, top):
    return sum(len(set(a["idx"].tolist()) & set(b["idx"].tolist())) for a, b in zip(got, exact)) / float(max(1, len(exact)) * top)



def _with_vectors(ctx, graph, scorer, raw, top, ef, n_gt, exact, reps=3):
    """GraphLayers::search_with_vectors (inline storage): the walk steered by the quantized scorer, every popped candidate scored on its original
    vector, the best base scores returned - rescoring fused into the walk (qmx_hnsw_search_with_vectors)."""
    try:
        res = graph.search_with_vectors(top, ef, scorer, raw)                     # warm-up
        t0 = time.perf_counter()
        for _ in range(reps):
            (out_raw, counts_raw), scored = graph.search_with_vectors(top, ef, scorer, raw, with_scored=True, raw_output=True)
        wall = (time.perf_counter() - t0) / reps
        res = [out_raw[i, :counts_raw[i]].copy() for i in range(n_gt)]
        return {"wall_ms_per_search": round(wall * 1e3, 3), "qps_wall": round(scorer.nq / wall, 1), "link_plus_base_vectors_scored_per_query": round(scored / scorer.nq, 1),
                "recall_at_10_vs_exact": round(_recall(res, exact, top), 4)}
    except Exception as e:
        return {"error": repr(e)[:300]}


def _timed_quantized(ctx, scorer, raw, top, oversampling, rescore, graph, ef, reps, row_bytes, n_rows_scanned=None):
    """reps calls of qmx_search_quantized; kernel time = HIP events around the quantized stage's scoring kernel (scan or walk)."""
    lib, F, qa = ctx["lib"], ctx["F"], ctx["qa"]
    F.check(lib.qmx_query_set_timing(scorer._h, 1))
    cnt = F.Counters()
    qa.search_quantized(scorer, raw, top, oversampling=oversampling, rescore=rescore, graph=graph, hnsw_ef=ef)          # warm-up
    ms, nl = C.c_float(), C.c_uint32()
    F.check(lib.qmx_query_timing(scorer._h, C.byref(ms), C.byref(nl)))
    scored = 0
    t0 = time.perf_counter()
    for _ in range(reps):   # (raw_output: the call as the shim would make it - host arrays in, host arrays out -, not 8 192 numpy slices per call)
        out_raw, counts_raw = qa.search_quantized(scorer, raw, top, oversampling=oversampling, rescore=rescore, graph=graph, hnsw_ef=ef, counters=cnt, raw_output=True)
        scored += int(cnt.vectors_scored)
    wall = (time.perf_counter() - t0) / reps
    res = [out_raw[i, :counts_raw[i]].copy() for i in range(scorer.nq)]
    F.check(lib.qmx_query_timing(scorer._h, C.byref(ms), C.byref(nl)))
    launches = max(1, int(nl.value))
    kernel_ms = ms.value / launches                        # per launch of the quantized stage's kernel
    per_launch_rows = (n_rows_scanned if n_rows_scanned is not None else scored / float(launches))
    gbps = per_launch_rows * row_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    sym = F.last_kernel(scorer._h)
    roof = _attach_traffic({"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                            "algorithmic_bytes_per_launch": int(per_launch_rows * row_bytes), "bytes_per_scored_row": row_bytes, "traffic": None}, sym,
                           n_rows_scanned if n_rows_scanned is not None else ctx["args"].config_rows or ctx["args"].rows)
    return res, {"kernel": sym, "kernel_ms": round(kernel_ms, 4), "launches_per_search": launches / float(reps),
                 "wall_ms_per_search": round(wall * 1e3, 3), "qps_wall": round(scorer.nq / wall, 1), "roofline": roof}, scored / float(reps)


def c3_section(ctx, rows):
    """BASELINE.json configs[2]: 10 M x 768 SQ-int8, dot; brute force + HNSW rescoring."""
    args, dev, lib, F, qa, np, torch = (ctx[k] for k in ("args", "dev", "lib", "F", "qa", "np", "torch"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dim, top = 768, 10
    n = args.config_rows or args.rows
    seed = 0x5EED0003
    t0 = time.perf_counter()
    if rows.shape != (n, dim):
        del rows
        torch.cuda.empty_cache()
        rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
    _latent(ctx, seed, 0, n, dim, out=rows)                 # refills the C2 block in place (30.72 GB, adopted, never copied)
    nq_h = args.hnsw_queries
    queries = _latent(ctx, seed, QUERY_ROW0, max(nq_h, 256), dim)
    torch.cuda.synchronize(dev)
    t_data = time.perf_counter() - t0
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    # ---- quantize: quantile = None fit (global min / max), encode on the device ----
    t0 = time.perf_counter()
    quant = qa.ScalarQuantizer.fit(rows, dim, qa.Distance.Dot)
    p = quant.params()
    codes = torch.empty((n, quant.quantized_vector_size()), dtype=torch.uint8, device=dev)
    F.check(lib.qmx_sq_encode(dev.index or 0, int(qa.Distance.Dot), C.byref(p), F.ptr(rows), n, dim, F.ptr(codes)))
    torch.cuda.synchronize(dev)
    t_enc = time.perf_counter() - t0
    enc = qa.EncodedVectorsU8(codes, quant)
    S = min(n, 200_000)
    host_codes_sample = codes[:S].cpu().numpy()
    host_rows_sample = rows[:2000].cpu().numpy()
    keep_codes = codes if args.verify else None
    del codes
    row_bytes = quant.quantized_vector_size()               # 772 B: SURVEY 8d
    out = {"workload": "C3: %s x d=768 SQ-int8 (min/max fit), dot; rows of low intrinsic dimension (32 latent coordinates + noise), cosine-normalised" % _human(n),
           "rows": n, "dim": dim, "row_bytes": row_bytes, "data_s": round(t_data, 2), "sq_fit_and_encode_s": round(t_enc, 3)}
    # ---- exact ground truth on the device (f32 brute force) ----
    n_gt = 256
    exact = qa.BatchFilteredSearcher(queries[:n_gt].cpu().numpy(), vs, top).peek_top_all()
    # ---- brute force over the codes, oversampling 2 + rescoring (PlainVectorIndex::search with quantization) ----
    bf = {}
    for Qb in (1, 32):
        nb = min(n_gt // Qb, 8)
        recs, stats = [], None
        for b in range(nb):
            qb = queries[b * Qb:(b + 1) * Qb].contiguous()
            scorer, raw = qa.new_raw_scorer(qb, enc), qa.new_raw_scorer(qb, vs)
            res, st, _ = _timed_quantized(ctx, scorer, raw, top, 2.0, True, None, 0, 5 if b == 0 else 1, row_bytes, n_rows_scanned=n)
            stats = stats or st
            recs.append(_recall(res, exact[b * Qb:(b + 1) * Qb], top))
        stats["recall_at_10_vs_exact"] = round(float(np.mean(recs)), 4)
        stats["queries_checked"] = nb * Qb
        bf["Q%d" % Qb] = stats
    out["brute_force_oversampling2_rescore"] = bf
    # ---- in-run oracle check: top-k over a sample of the codes, bit-exact scores; encoded rows byte-exact ----
    if args.verify:
        import oracle_ffi as O
        osq = O.SqOracle(O.DOT, dim, quant.alpha, quant.offset)
        enc_ok = bool(np.array_equal(osq.encode_rows(host_rows_sample), host_codes_sample[:2000]))
        osq.rows = host_codes_sample
        qpre = queries[:2].cpu().numpy()
        ids = np.arange(S, dtype=np.uint32)
        got = qa.BatchFilteredSearcher(qpre, vs, top, quantized_vectors=enc).peek_top_iter(ids)
        sc = osq.score_points(qpre, ids)
        top_ok = all(np.array_equal(np.sort(sc[i])[::-1][:top].view(np.uint32), got[i]["score"].view(np.uint32)) for i in range(2))
        out["oracle_check"] = {"encoded_rows_byte_exact_first_2000": enc_ok, "topk_scores_bit_exact_on_%dk_sample" % (S // 1000): bool(top_ok)}
    # ---- HNSW: build THROUGH the SQ scorer (hnsw/build.rs:334-341), SQ walk, oversampling 2, f32 rescoring ----
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    graph = qa.GraphLayers.build(enc, m=16, ef_construct=100, seed=42)
    t_build = time.perf_counter() - t0
    qh_all = queries[:nq_h].contiguous()
    scorer, raw = qa.new_raw_scorer(qh_all, enc), qa.new_raw_scorer(qh_all, vs)
    res, st, scored = _timed_quantized(ctx, scorer, raw, top, 2.0, True, graph, 128, 3, row_bytes)
    st.update({"m": 16, "ef_construct": 100, "ef": 128, "oversampling": 2.0, "searches_per_launch": nq_h,
               "build_s": round(t_build, 2), "build_points_per_s": round(n / t_build, 1),
               "points_scored_per_query": round(scored / nq_h, 1), "recall_at_10_vs_exact": round(_recall(res[:n_gt], exact, top), 4)})
    st["search_with_vectors_ef128"] = _with_vectors(ctx, graph, scorer, raw, top, 128, n_gt, exact)
    # recall-vs-ef of the same graph, f32 walk (graph quality without the quantizer)
    st["recall_f32_walk_vs_ef"] = {str(ef): round(_recall(graph.search(top, ef, qa.new_raw_scorer(queries[:n_gt].contiguous(), vs)), exact, top), 4)
                                   for ef in (64, 128, 256)}
    if args.verify:
        # the CPU oracle walks THE SAME graph with its SQ scorer: identical ids and score bits expected (host copy of the codes + links)
        import oracle_ffi as O
        try:
            t0 = time.perf_counter()
            osq_all = O.SqOracle(O.DOT, dim, quant.alpha, quant.offset)
            osq_all.rows = keep_codes.cpu().numpy()
            walker = O.Hnsw.from_plain(graph.export_plain(), n)
            flags = O.DenseStorage(O.F32, O.DOT, np.zeros((1, dim), dtype=np.float32))
            flags.st.n = n
            nchk = min(256, int(queries.shape[0]))      # (16 searches were thin evidence at 10 M rows)
            want = walker.search_sq(flags, osq_all, queries[:nchk].cpu().numpy(), 2 * top, 128)
            got = graph.search(2 * top, 128, qa.new_raw_scorer(queries[:nchk].contiguous(), enc))
            st["oracle_walk_check"] = {"same_ids": "%d/%d" % (sum(int(a["idx"].tolist() == b["idx"].tolist()) for a, b in zip(got, want)), nchk),
                                       "same_score_bits": "%d/%d" % (sum(int(np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32)))
                                                                         for a, b in zip(got, want)), nchk),
                                       "seconds": round(time.perf_counter() - t0, 1)}
            del osq_all, walker
        except Exception as e:
            st["oracle_walk_check"] = {"error": repr(e)[:300]}
    out["hnsw_sq_walk_rescore"] = st
    del keep_codes, graph, enc, vs
    return out, rows


def tq_section(ctx, rows):
    """The C3 rows quantized with TurboQuant (4 bits, TQMode::Normal, dot): device encode, brute force with oversampling 2 + f32 rescoring."""
    args, dev, lib, F, qa, np, torch = (ctx[k] for k in ("args", "dev", "lib", "F", "qa", "np", "torch"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    n, dim = rows.shape
    top, seed = 10, 0x5EED0003
    queries = _latent(ctx, seed, QUERY_ROW0, 256, dim)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    quant = qa.TurboQuantizer(dim, qa.Distance.Dot, 0)
    p = quant.params()
    row_bytes = quant.quantized_vector_size()
    codes = torch.empty((n, row_bytes), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    F.check(lib.qmx_tq_encode(dev.index or 0, int(qa.Distance.Dot), dim, C.byref(p), F.ptr(rows), n, F.ptr(codes)))
    torch.cuda.synchronize(dev)
    t_enc = time.perf_counter() - t0
    enc = qa.EncodedVectorsTQ(codes, quant)
    S = min(n, 20_000)
    host_codes_sample = codes[:S].cpu().numpy()
    host_rows_sample = rows[:300].cpu().numpy()
    del codes
    out = {"workload": "the rows of C3 as TurboQuant 4-bit (Hadamard rotation + Lloyd-Max codebook), dot: %s x d=%d" % (_human(n), dim),
           "rows": n, "dim": dim, "row_bytes": row_bytes, "tq_encode_s": round(t_enc, 3)}
    n_gt = 256
    exact = qa.BatchFilteredSearcher(queries[:n_gt].cpu().numpy(), vs, top).peek_top_all()
    bf = {}
    for Qb in (1, 32):
        nb = min(n_gt // Qb, 8)
        recs, stats = [], None
        for b in range(nb):
            qb = queries[b * Qb:(b + 1) * Qb].contiguous()
            scorer, raw = qa.new_raw_scorer(qb, enc), qa.new_raw_scorer(qb, vs)
            res, st, _ = _timed_quantized(ctx, scorer, raw, top, 2.0, True, None, 0, 5 if b == 0 else 1, row_bytes, n_rows_scanned=n)
            stats = stats or st
            recs.append(_recall(res, exact[b * Qb:(b + 1) * Qb], top))
        stats["recall_at_10_vs_exact"] = round(float(np.mean(recs)), 4)
        stats["queries_checked"] = nb * Qb
        bf["Q%d" % Qb] = stats
    out["brute_force_oversampling2_rescore"] = bf
    if args.verify:
        import oracle_ffi as O
        otq = O.TqOracle(O.DOT, dim, O.TQ_BITS4)
        enc_ok = bool(np.array_equal(otq.encode_rows(host_rows_sample), host_codes_sample[:300]))
        otq.rows = host_codes_sample
        qpre = queries[:2].cpu().numpy()
        ids = np.arange(S, dtype=np.uint32)
        got = qa.BatchFilteredSearcher(qpre, vs, top, quantized_vectors=enc).peek_top_iter(ids)
        sc = otq.score_points(qpre, ids)
        top_ok = all(np.array_equal(np.sort(sc[i])[::-1][:top].view(np.uint32), got[i]["score"].view(np.uint32)) for i in range(2))
        out["oracle_check"] = {"encoded_rows_byte_exact_first_300": enc_ok, "topk_scores_bit_exact_on_%dk_sample" % (S // 1000): bool(top_ok)}
    del enc, vs
    return out


def c4_section(ctx):
    """BASELINE.json configs[3]: 10 M x 1536, PQ m = 96 (8-bit), HNSW ef = 128, LUT on the matrix cores."""
    args, dev, lib, F, qa, np, torch = (ctx[k] for k in ("args", "dev", "lib", "F", "qa", "np", "torch"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dim, chunk, top = 1536, 16, 10
    n = args.config_rows or args.rows
    seed = 0x5EED0004
    t0 = time.perf_counter()
    rows = _latent(ctx, seed, 0, n, dim)                    # 61.4 GB at 10 M rows
    nq_h = args.hnsw_queries
    queries = _latent(ctx, seed, QUERY_ROW0, max(nq_h, 256), dim)
    torch.cuda.synchronize(dev)
    t_data = time.perf_counter() - t0
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    # ---- codebook: kmeans.rs on a 10 000-row sample (KMEANS_SAMPLE_SIZE), on the device; encode on the device ----
    t0 = time.perf_counter()
    stride = max(1, n // 10000)
    sample = rows[::stride][:10000].contiguous()
    cen = torch.zeros((256, dim), dtype=torch.float32, device=dev)
    iters = np.zeros(dim // chunk, dtype=np.uint32)
    F.check(lib.qmx_pq_train(dev.index or 0, F.ptr(sample), sample.shape[0], dim, chunk, 256, 100, 1e-5, 1, F.ptr(cen), F.ptr(iters)))
    torch.cuda.synchronize(dev)
    t_train = time.perf_counter() - t0
    cen_h = cen.cpu().numpy()
    quant = qa.ProductQuantizer(dim, qa.Distance.Dot, chunk, cen_h, lut_mfma=True)    # north_star: PQ LUT build via MFMA
    p = quant.params()
    codes = torch.empty((n, quant.m), dtype=torch.uint8, device=dev)
    t0 = time.perf_counter()
    F.check(lib.qmx_pq_encode(dev.index or 0, C.byref(p), F.ptr(rows), n, dim, F.ptr(codes)))
    torch.cuda.synchronize(dev)
    t_enc = time.perf_counter() - t0
    enc = qa.EncodedVectorsPQ(codes, quant)
    out = {"workload": "C4: %s x d=1536 PQ m=%d (chunk 16, 256 centroids), dot; rows of low intrinsic dimension, cosine-normalised; LUT via v_mfma_f32_32x32x2_f32"
                       % (_human(n), quant.m), "rows": n, "dim": dim, "row_bytes": quant.m, "data_s": round(t_data, 2),
           "pq_kmeans_train_s": round(t_train, 3), "kmeans_iterations_max": int(iters.max()), "pq_encode_s": round(t_enc, 3)}
    n_gt = 256
    exact = qa.BatchFilteredSearcher(queries[:n_gt].cpu().numpy(), vs, top).peek_top_all()
    # ---- LUT build on the matrix cores: time of encode_query for a batch (qmx_query_update), its flops against the f32 MFMA peak ----
    qh_all = queries[:nq_h].contiguous()
    scorer = qa.new_raw_scorer(qh_all, enc)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(3):
        F.check(lib.qmx_query_update(scorer._h, F.ptr(qh_all)))
    F.check(lib.qmx_query_synchronize(scorer._h))
    lut_ms = (time.perf_counter() - t0) / 3 * 1e3
    lut_flops = 2.0 * 256 * dim * nq_h                       # 2 x 256 x d flop per query
    out["lut_build_mfma"] = {"queries": nq_h, "ms_incl_preprocess": round(lut_ms, 3), "flops": lut_flops,
                             "roofline": {"bound": "mfma", "achieved": round(lut_flops / (lut_ms * 1e-3) / 1e12, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": round(lut_flops / (lut_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                          "note": "wall time of qmx_query_update (cosine preprocess + LUT + the 96 KiB/query LUT write: %d MB): write-bound, not MFMA-bound" % (nq_h * 96 // 1024)}}
    # ---- HNSW: build through the PQ scorer (point_scorer.rs:197-212), PQ walk ef = 128 ----
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    graph = qa.GraphLayers.build(enc, m=16, ef_construct=100, seed=42, original=vs)
    t_build = time.perf_counter() - t0
    raw = qa.new_raw_scorer(qh_all, vs)
    walks = {}
    for name, over, resc in (("no_rescoring", 0.0, False), ("oversampling2_rescore", 2.0, True), ("oversampling4_rescore", 4.0, True)):
        res, st, scored = _timed_quantized(ctx, scorer, raw, top, over, resc, graph, 128, 3, quant.m)
        st.update({"points_scored_per_query": round(scored / nq_h, 1), "recall_at_10_vs_exact": round(_recall(res[:n_gt], exact, top), 4)})
        walks[name] = st
    # the same walk without LUTs (option hnsw_pq_direct_walk, pq.hip HopPQDirect: every LUT entry recomputed from the codebook - the exact LUT's bits,
    # 1 / 20 of the HBM traffic, profiles/r4_pq_direct_walk.md); timed beside the default so that the driver's line carries both at full size
    qa.set_option("hnsw_pq_direct_walk", 1)
    try:
        res_d, st_d, scored_d = _timed_quantized(ctx, scorer, raw, top, 0.0, False, graph, 128, 3, quant.m)
        st_d.update({"points_scored_per_query": round(scored_d / nq_h, 1), "recall_at_10_vs_exact": round(_recall(res_d[:n_gt], exact, top), 4)})
        walks["no_rescoring_lut_free_walk"] = st_d
    except Exception as e:
        walks["no_rescoring_lut_free_walk"] = {"error": repr(e)[:200]}
    finally:
        qa.set_option("hnsw_pq_direct_walk", -1)
    hn = {"m": 16, "ef_construct": 100, "ef": 128, "searches_per_launch": nq_h, "build_through": "PQ scorer (LUT of the original vector per insertion, score_internal for the heuristic)",
          "build_s": round(t_build, 2), "build_points_per_s": round(n / t_build, 1), "walks": walks}
    hn["search_with_vectors_ef128"] = _with_vectors(ctx, graph, scorer, raw, top, 128, n_gt, exact)
    hn["recall_f32_walk_vs_ef"] = {str(ef): round(_recall(graph.search(top, ef, qa.new_raw_scorer(queries[:n_gt].contiguous(), vs)), exact, top), 4)
                                   for ef in (64, 128, 256)}
    # brute force over the codes for reference (what the quantizer alone can do on these rows)
    bfs = qa.new_raw_scorer(queries[:32].contiguous(), enc)
    bfr = qa.new_raw_scorer(queries[:32].contiguous(), vs)
    res, st, _ = _timed_quantized(ctx, bfs, bfr, top, 2.0, True, None, 0, 3, quant.m, n_rows_scanned=n)
    st["recall_at_10_vs_exact"] = round(_recall(res, exact[:32], top), 4)
    out["brute_force_Q32_oversampling2_rescore"] = st
    if args.verify:
        import oracle_ffi as O
        try:
            t0 = time.perf_counter()
            opq = O.PqOracle(O.DOT, dim, chunk, cen_h)
            host_codes = codes.cpu().numpy()
            enc_ok = bool(np.array_equal(opq.encode(rows[:1000].cpu().numpy()), host_codes[:1000]))
            opq.codes = host_codes
            walker = O.Hnsw.from_plain(graph.export_plain(), n)
            flags = O.DenseStorage(O.F32, O.DOT, np.zeros((1, dim), dtype=np.float32))
            flags.st.n = n
            nchk = min(256, int(queries.shape[0]))      # (16 searches were thin evidence at 10 M rows)
            qpre = queries[:nchk].cpu().numpy()
            # the oracle's LUT is the exact-order one; the device walk under test uses the MFMA LUT (<= 1e-5): compare against a device walk
            # with the exact-order LUT for bits, and report how the MFMA-LUT walk compares
            quant_exact = qa.ProductQuantizer(dim, qa.Distance.Dot, chunk, cen_h, lut_mfma=False)
            enc_exact = qa.EncodedVectorsPQ(codes, quant_exact)
            want = walker.search_pq(flags, opq, qpre, 2 * top, 128)
            got = graph.search(2 * top, 128, qa.new_raw_scorer(queries[:nchk].contiguous(), enc_exact))
            got_mfma = graph.search(2 * top, 128, qa.new_raw_scorer(queries[:nchk].contiguous(), enc))
            hn["oracle_walk_check"] = {"codes_byte_exact_first_1000": enc_ok,
                                       "exact_lut_same_ids": "%d/%d" % (sum(int(a["idx"].tolist() == b["idx"].tolist()) for a, b in zip(got, want)), nchk),
                                       "exact_lut_same_score_bits": "%d/%d" % (sum(int(np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32)))
                                                                                   for a, b in zip(got, want)), nchk),
                                       "mfma_lut_same_id_sets": "%d/%d" % (sum(int(set(a["idx"].tolist()) == set(b["idx"].tolist())) for a, b in zip(got_mfma, want)), nchk),
                                       "mfma_lut_max_rel_score_err": float(max(np.max(np.abs(a["score"][:min(len(a), len(b))] - b["score"][:min(len(a), len(b))]) /
                                                                                      np.maximum(np.abs(b["score"][:min(len(a), len(b))]), 1e-30)) for a, b in zip(got_mfma, want))),
                                       "seconds": round(time.perf_counter() - t0, 1)}
        except Exception as e:
            hn["oracle_walk_check"] = {"error": repr(e)[:300]}
    out["hnsw_pq_walk"] = hn
    return out


if __name__ == "__main__":
    main()
