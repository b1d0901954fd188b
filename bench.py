#!/usr/bin/env python3
"""bench.py — QPS of brute-force top-10 search on the MI355X scorer (BASELINE.json metric), + the other single-GPU configs.

Timed region (the headline's `value`).  Workload at N=1 = BASELINE.json configs[1] ("C2"): one segment of 10 M x d=768 f32, cosine,
brute-force exact top-10, resident in HBM.  A *step* = one pass of the hot path over one batch of Q queries: Metric::preprocess of
the batch (qmx_query_update) + one scan of the whole segment with per-query top-k (qmx_search_topk_async =
BatchFilteredSearcher::peek_top_iter) [+ for N>1 the RCCL all-gather of the per-GPU top-k and the k-way merge].  1024 distinct
queries are cycled in batches.  Before the W warm-up steps an untimed steady-state run of --prewarm-ms (150 ms, `config.prewarm_steps`) belongs to the set-up,
like generating the block.  The harness whose shape this replaces: /root/reference/lib/segment/benches/vector_search.rs:21,34-104.

N>1: one rank per GPU.  Under torchrun (RANK / WORLD_SIZE set) the ranks are the launcher's; WITHOUT it `--gpus N` launches its own ranks
(re-exec under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`), and refuses (exit 2) when fewer than
N devices are visible or when WORLD_SIZE disagrees with --gpus: a run never reports fewer ranks than it was asked for.
  --scaling weak (default) = configs[4] ("C5"): rank r holds its own 10 M-row segment (seed + r), every rank scores the same query batch
      against its segment, the per-rank answers (one packed record: Q x 10 x 8 B of lists + Q x 4 B of counts) are all-gathered over RCCL/xGMI in
      ONE collective per step and merged (BatchResultAggregator semantics).  Per-GPU work is fixed; value = Q * steps / time = queries per second
      against the whole collection of N x 10 M vectors (ideal weak scaling: value(N) == value(1)); config.segment_searches_per_s = N x value.
      N > 1 runs the experiment of N = 1: the same --in-flight batches on their own streams (steps alternate between them in the same order on
      every rank), + one collective and one merge per step; config.per_step_us reports local search / all-gather / merge from stream events.
  --scaling strong: ONE 10 M-row segment row-split over the ranks, same gather + merge; value = Q * steps / time.
  Rank 0's side measurements (block-stream roofline, one-process fan-out) run AFTER the process group has ended: no rank waits in an RCCL barrier meanwhile.

Output.  The LAST stdout line is the compact headline (<= 4 KB, `headline()`): metric / value / ms_per_step / config / `roofline` /
`cpu_baseline` + a few numbers of every secondary leg.  Everything else (`batch_sweep`, `robustness`, `one_process_fanout`, the C3 / TQ4 / C4
legs in full) is written to bench_details.json (--details) and echoed on stderr.

`roofline` (N=1) carries BOTH fractions under explicit names, measured in the same run with HIP-event pairs on the kernel's own stream
(qmx_query_set_timing / qmx_query_timing):
  block_stream  SURVEY 8(d) / north_star's figure: the exact scan streaming the stored f32 block itself (rows x 3072 B = 30.72 GB) once for 16
                queries (scan_f32_mfma16_kernel) - and once for 1 query (scan_kernel<RowF32>) in block_stream_q1.  The top-level
                bound / achieved / peak / frac / traffic are this point's.
  timed_kernel  the dominant kernel of the timed step on the bytes IT streams: by default the prefilter over the derived int8 copy (768 B per
                row, two launches per pass) whose survivors are re-scored exactly in f32 - the lists are the exact scan's, checked in the run.
`cpu_baseline` = the CPU oracle (AVX2+FMA restatement of the reference's scorer and its peek_top_iter loop) timed on this box's host cores on a
bounded sample of the same rows.  The oracle is the checker: nothing in the timed region touches it.
"""
import argparse
import ctypes as C
import importlib
import json
import math
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from bench_roofline import (HBM_PEAK_GBPS, MFMA_F16_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS, MFMA_I8_PEAK_TOPS,  # noqa: E402,F401
                            _attach_traffic, _human, _pmc_entry, _pmc_traffic, _roofline, _same_kernel, _stddev, _timed_path)

HEADLINE_MAX_BYTES = 4096
BLOCK_STREAM_BATCH = 16


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=128, help="queries per step (Q); more than 64 run 128 per pass through the prefilter + exact verification")
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
    ap.add_argument("--no-hbm-point", action="store_true", help="skip the block-stream points (the exact scan over the f32 block at 16 queries and at 1)")
    ap.add_argument("--no-other-copy-point", action="store_true",
                    help="skip the secondary measurement of the same search over the OTHER derived copy (the f16 half copy when the int8 copy is timed, and vice versa)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the Q in {1, 8, 32, 128} x {exact, prefilter} sweep")
    ap.add_argument("--no-robustness", action="store_true", help="skip the C2 search on latent / duplicated rows (needs 46 GB more HBM)")
    ap.add_argument("--verify", type=int, default=1, help="check samples against the oracle (C2 first batch, C3 / C4 scans and walks)")
    ap.add_argument("--configs", default="c1,c3,tq,c4", help="comma list of the secondary single-GPU configs to run (empty = none)")
    ap.add_argument("--config-rows", type=int, default=0, help="rows of C3 / C4 (0 = --rows)")
    ap.add_argument("--iid-walk", action="store_true",
                    help="also run the C3 walk on C2's own iid rows (SURVEY 8d's C3 rows): one more 10 M-point graph build, +60 s - off by default to keep the default "
                         "command near four minutes; the round's record of it: profiles/r6_bench_with_iid_walk_details.json")
    ap.add_argument("--hnsw-queries", type=int, default=8192, help="searches per launch of the HNSW walks")
    ap.add_argument("--in-flight", type=int, default=4, choices=[1, 2, 3, 4, 5, 6, 8],
                    help="query batches in flight on one GPU: consecutive steps take turns on this many query handles, each with its own stream, so that one batch's "
                         "head (preprocess, sample pre-scan, pack), the gap between its two scans (probe, exact gather, bound) and its tail (verification, sort) run "
                         "beside or between the other batches' scans; 1 = one handle, one stream.  (Rounds 3 - 5: 2.  Round 6: the int8 scan leaves 16 KiB of LDS per CU "
                         "free, the small kernels run beside it, and four batches keep a scan ready at every hand-over: 87.5 k -> 91.4 k QPS, profiles/r6_i8_lanes_experiment.txt)")
    ap.add_argument("--fanout-rows", type=int, default=1_000_000,
                    help="rows per segment of the one-process fan-out legs (qmx_sharded_hnsw_build + qmx_sharded_search_topk over one segment per device); 0 = skip")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed steady-state run BEFORE the --warmup steps (part of the set-up, like generating the block): steps are issued for this many "
                         "milliseconds so that clocks, both batches in flight and the allocator are where a serving process has them; reported in config.prewarm_steps")
    ap.add_argument("--details", default=os.path.join(ROOT, "bench_details.json"), help="where the full result (every leg) is written")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="collective backend of N>1 (nccl = RCCL; gloo only with --test-backend)")
    ap.add_argument("--ranks-share-gpu", action="store_true",
                    help="TESTING the N > 1 path on a box with fewer GPUs than ranks: rank r runs on device r %% (visible devices), the collective is gloo over "
                         "device tensors (--backend gloo).  Everything of the N > 1 experiment but RCCL itself runs - lanes, streams, the packed all-gather "
                         "and merge on the device, the side legs after the process group - and the line says that it is no scaling measurement")
    ap.add_argument("--test-backend", default="",
                    help="module:factory of an injected compute backend (tests only: the launcher, the collective and the line on CPU; never a measurement)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------------
# launcher: --gpus N starts N ranks itself when no launcher did
# ------------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(args, argv, port=None):
    """The command `--gpus N` re-executes itself under when no launcher set WORLD_SIZE (segments_searcher.rs:250-285 fans out inside one process;
    here one process per GPU, as gpu_devices_manager.rs:120-143 hands one device to one build)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
            "--master-port", str(port or _free_port()), os.path.abspath(__file__)] + list(argv)


def resolve_world(args, argv, environ=None, device_count=None):
    """-> ("run", rank, local_rank, world) | ("exec", cmd) | ("fail", message).  Pure: the caller acts."""
    environ = os.environ if environ is None else environ
    if args.gpus < 1:
        return ("fail", "--gpus must be >= 1")
    if args.backend == "gloo" and not args.test_backend and not args.ranks_share_gpu:
        return ("fail", "--backend gloo needs --test-backend (the product has no CPU path) or --ranks-share-gpu (device tensors over gloo: a test of the N > 1 path)")
    if args.ranks_share_gpu and (args.backend != "gloo" or args.test_backend):
        return ("fail", "--ranks-share-gpu goes with --backend gloo (RCCL refuses two ranks on one device) and the real compute backend")
    if args.ranks_share_gpu:
        device_count = None      # (any number of visible devices serves any number of ranks)
    if "WORLD_SIZE" in environ:
        world = int(environ["WORLD_SIZE"])
        if world != args.gpus:
            return ("fail", "--gpus %d but the launcher started WORLD_SIZE=%d ranks: refusing to report a different device count than asked" % (args.gpus, world))
        if not args.test_backend and device_count is not None and device_count < int(environ.get("LOCAL_WORLD_SIZE", world)):
            return ("fail", "--gpus %d but only %d device(s) visible" % (args.gpus, device_count))
        return ("run", int(environ.get("RANK", "0")), int(environ.get("LOCAL_RANK", "0")), world)
    if args.gpus == 1:
        return ("run", 0, 0, 1)
    if not args.test_backend and device_count is not None and device_count < args.gpus:
        return ("fail", "--gpus %d but only %d device(s) visible: refusing to run on fewer GPUs than asked" % (args.gpus, device_count))
    return ("exec", launch_command(args, argv))


# ------------------------------------------------------------------------------------------------------------------------
# the compact headline (last stdout line)
# ------------------------------------------------------------------------------------------------------------------------
def _short(sym, limit=72):
    """kernel symbol without 'void', the qmx:: prefixes and the argument list"""
    s = str(sym).replace("void ", "").replace("qmx::", "")
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            cut = i
            break
    return s[:cut].replace(", ", ",")[:limit]


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


LEG_COLUMNS = "legs[kernel,ms,bound,frac,qps,recall@10]"


def _qps(v):
    """whole numbers from 1 000 QPS up (the headline is 4 KB)"""
    return int(round(v)) if isinstance(v, (int, float)) and v >= 1000 else v


def _leg(st):
    """[kernel, kernel_ms, bound, frac, qps_wall, recall@10] of one secondary leg (LEG_COLUMNS)"""
    if not isinstance(st, dict):
        return None
    if "error" in st:
        return "error: " + str(st["error"])[:60]
    r = st.get("roofline", {})
    bound, frac = r.get("bound"), r.get("frac")
    if "lds" in r:                      # a kernel priced against the LDS issue roof carries that fraction (its HBM fraction stays in the details)
        bound, frac = "lds", r["lds"].get("frac")
    if isinstance(st.get("matrix_cores"), dict):      # ... against the matrix cores' peak likewise (the 128-query TurboQuant pass)
        bound, frac = "mfma", st["matrix_cores"].get("frac")
    ms = st.get("kernel_ms")
    return [_short(st.get("kernel", "")).replace("_kernel", "").replace("hnsw_search<", "walk<")[:17], round(ms, 3) if isinstance(ms, float) else ms, bound, frac, _qps(st.get("qps_wall")), st.get("recall_at_10_vs_exact")]


def _block_stream(p):
    if not isinstance(p, dict) or "error" in p or "kernel_ms" not in p:
        return None
    return {"kernel": _short(p["kernel"]), "batch": p["batch"], "kernel_ms": p["kernel_ms"], "launches": p.get("launches_timed"),
            "algorithmic_bytes": p["algorithmic_bytes_per_launch"], "frac": p["frac"], "traffic_over_algorithmic": p.get("traffic_over_algorithmic")}


def compact_roofline(result):
    """`roofline` of the headline: top level = the SURVEY 8(d) block stream (the f32 block read once for 16 queries) when it was measured in this
    run, else the timed kernel; both under explicit names."""
    r = result.get("roofline") or {}
    timed = {"kernel": _short(r.get("kernel", "")), "kernel_ms": r.get("kernel_ms"), "launches": r.get("launches_timed"),
             "launches_per_pass": r.get("launches_per_pass"), "bytes_streamed": r.get("algorithmic_bytes_per_launch"),
             "GBps": (r.get("hbm") or {}).get("achieved_GBps"), "frac": (r.get("hbm") or {}).get("frac"),
             "kernel_ms_in_timed_region": r.get("kernel_ms_in_timed_region"),
             "traffic_over_bytes": r.get("traffic_over_algorithmic"),
             "mfma_frac": (r.get("mfma") or {}).get("frac")}
    pf = r.get("prefilter_per_batch")
    if isinstance(pf, dict) and "error" not in pf:
        timed["verified_rows_per_query"] = pf.get("verified_rows_per_query")
        timed["fallback_queries"] = pf.get("fallback_queries")
    bs = _block_stream(result.get("roofline_hbm_point_q16"))
    bs1 = _block_stream(result.get("roofline_hbm_point_q1"))
    if bs is None:
        out = _pick(r, "bound", "achieved", "peak", "unit", "frac", "traffic")
        out.setdefault("traffic", None)
        out["of"] = "timed_kernel (no block-stream point in this run)"
    else:
        p = result["roofline_hbm_point_q16"]
        out = {"bound": "hbm", "achieved": p["achieved"], "peak": p["peak"], "unit": "GB/s", "frac": p["frac"], "traffic": p.get("traffic"),
               # (counters cannot be read from inside the process: `traffic` is this kernel symbol's entry of profiles/pmc_traffic.json - separate rocprofv3 --pmc
               # passes at this row count, the file names the round that measured it)
               "traffic_source": "lookup <- %s" % os.path.basename(str(p.get("traffic_source") or "?")),
               "of": "block_stream = SURVEY 8(d): exact scan of the stored f32 block; `value` is timed_kernel's path",
               "block_stream": bs}
        if bs1:
            out["block_stream_q1"] = _pick(bs1, "kernel", "batch", "kernel_ms", "frac", "traffic_over_algorithmic")
    out["timed_kernel"] = timed
    return out


def _walk_summary(ow, cost=None):
    """oracle_walk_check of a walk leg in three short strings"""
    if not isinstance(ow, dict) or not ow:
        return None
    if "error" in ow:
        return {"error": str(ow["error"])[:80]}
    g = lambda *ks: next((ow[k] for k in ks if k in ow), None)      # noqa: E731
    out = {"default_walk": "ids %s, bits %s" % (g("same_ids", "exact_lut_same_ids"), g("same_score_bits", "exact_lut_same_score_bits"))}
    if "tie_explained" in ow:
        out["tie_explained"] = ow["tie_explained"]
    if "unexplained" in ow:
        out["unexplained"] = len(ow["unexplained"])
    if "reference_heap_order_same_ids" in ow:
        out["reference_heap_order"] = "ids %s, bits %s, pops %s" % (ow["reference_heap_order_same_ids"], ow.get("reference_heap_order_same_score_bits"),
                                                                   ow.get("reference_heap_order_same_pops"))
    if isinstance(cost, dict) and "kernel_ms" in cost:
        out["reference_heap_order_ms"] = [cost["kernel_ms"], cost.get("over_default_walk")]      # [kernel ms per launch, x the default walk]
    return out


def _configs_summary(cfg):
    legs, out = {}, {LEG_COLUMNS: None}
    put = lambda name, st: legs.__setitem__(name, _leg(st)) if st is not None else None      # noqa: E731
    c1 = cfg.get("C1")
    if isinstance(c1, dict):
        if "error" in c1:
            out["C1"] = {"error": str(c1["error"])[:80]}
        else:
            cp = c1.get("cpu_oracle_port", {})
            out["C1"] = {"cpu_qps_1_thread": cp.get("one_thread", {}).get("qps"), "cpu_qps_all_cores": cp.get("all_cores", {}).get("qps"), "cores": cp.get("cores"),
                         "device_qps_Q32": c1.get("device", {}).get("Q32", {}).get("qps_wall_sync_per_batch"), "device_equals_oracle": c1.get("device_equals_oracle_bit_exact")}
    c3 = cfg.get("C3")
    if isinstance(c3, dict):
        if "error" in c3:
            out["C3"] = {"error": str(c3["error"])[:80]}
        else:
            h = c3.get("hnsw_sq_walk_rescore", {})
            bf = c3.get("brute_force_oversampling2_rescore", {})
            put("C3.walk_iid_rows", cfg.get("C3_walk_on_iid_rows"))
            put("C3.scan_Q1", bf.get("Q1"))
            put("C3.scan_Q32", bf.get("Q32"))
            put("C3.scan_Q128", bf.get("Q128"))
            put("C3.walk", h)
            ow = h.get("oracle_walk_check", {})
            out["C3"] = {"rows": "latent-32", "build_s": h.get("build_s"), "oracle_walk": _walk_summary(ow, h.get("reference_heap_order")),
                         "oracle_scan_ok": all(v is True for v in c3.get("oracle_check", {"-": None}).values())}
    tq = cfg.get("TQ4")
    if isinstance(tq, dict):
        if "error" in tq:
            out["TQ4"] = {"error": str(tq["error"])[:80]}
        else:
            for k, v in tq.get("brute_force_oversampling2_rescore", {}).items():
                put("TQ4.scan_" + k, v)
            out["TQ4"] = {"oracle_scan_ok": all(v is True for v in tq.get("oracle_check", {"-": None}).values())}
    c4 = cfg.get("C4")
    if isinstance(c4, dict):
        if "error" in c4:
            out["C4"] = {"error": str(c4["error"])[:80]}
        else:
            h = c4.get("hnsw_pq_walk", {})
            w = h.get("walks", {})
            lut = c4.get("lut_build_mfma", {})
            put("C4.walk", w.get("no_rescoring"))
            put("C4.scan_Q32", c4.get("brute_force_Q32_oversampling2_rescore"))
            ow = h.get("oracle_walk_check", {})
            out["C4"] = {"rows": "latent-32", "build_s": h.get("build_s"), "hop_prefilter": _pick(h.get("hop_prefilter", {}), "survivors_per_hop", "G_requests_per_s"), "lut_mfma": _pick(lut.get("kernel_roofline", lut.get("roofline", {})), "achieved", "frac", "kernel_ms"),
                         "oracle_walk": _walk_summary(ow, h.get("reference_heap_order"))}
    out[LEG_COLUMNS] = legs
    return out


def headline(result, details_path=None):
    """The compact line the driver parses: every contract key + roofline + cpu_baseline + a few numbers per secondary leg; <= HEADLINE_MAX_BYTES
    (optional groups are dropped, last first, if a run ever grows past it)."""
    h = _pick(result, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "value_stddev", "higher_is_better", "scaling")
    h["vs_baseline"] = result.get("vs_baseline")
    h.update(_pick(result, "dtype", "data", "rccl_ranks", "collective"))
    c = result.get("config", {})
    hc = _pick(c, "workload", "rows_per_gpu", "dim", "batch", "top", "batches_in_flight", "prewarm_steps")
    if result.get("n_gpus", 1) > 1:
        hc.update(_pick(c, "collection_qps", "segment_searches_per_s", "collectives_per_step", "per_step_us", "unit_of_value"))
    if "timed_path" in c and result.get("n_gpus", 1) > 1:      # (at N = 1 `dtype` and roofline.timed_kernel say it; the details carry the sentence)
        hc["timed_path"] = c["timed_path"].split(":")[0].replace(" of the block (1 B / element, int8 matrix cores)", "").replace(" of the survivors", "")[:100]
    dc = c.get("derived_copy")
    if isinstance(dc, dict):
        hc["derived_copy"] = "%s (requested %s)" % (dc.get("derived_copy"), dc.get("requested"))      # (the trial's timings: details)
    h["config"] = hc
    h["roofline"] = compact_roofline(result)
    cb = result.get("cpu_baseline")
    if isinstance(cb, dict):
        h["cpu_baseline"] = _pick(cb, "value", "unit", "cores", "kind")
        h["cpu_baseline"]["sample"] = str(cb.get("sample_short", cb.get("sample", "")))[:64]
        if "single_thread" in cb:
            h["cpu_baseline"]["single_thread_value"] = cb["single_thread"].get("value")
        h["cpu_baseline"]["gpu_matches_oracle_on_sample_bit_exact"] = cb.get("gpu_matches_oracle_on_sample_bit_exact")
    h["checks"] = _pick(result, "prefilter_equals_exact_scan_whole_block", "recall_at_10")
    optional = []
    sw = result.get("batch_sweep")
    if isinstance(sw, dict):
        h["batch_sweep"] = {"[qps,frac]": {k: ([v.get("qps"), v.get("frac")] if "error" not in v else "error") for k, v in sw.items()}}
        optional.append("batch_sweep")
    rb = result.get("robustness")
    if isinstance(rb, dict):
        if "error" in rb:
            h["robustness"] = {"error": str(rb["error"])[:80]}
        else:
            legs = []
            for v in rb.values():
                legs += [v] if ("qps" in v or "error" in v) else [vv for vv in v.values() if isinstance(vv, dict)]
            ok = [g for g in legs if "qps" in g]
            h["robustness"] = {"legs": len(legs), "failed": len(legs) - len(ok), "every_list_equals_exact_scan": all(g.get("equals_exact_scan_whole_block") for g in ok),
                               "qps_min": min([g["qps"] for g in ok] or [None]), "qps_max": max([g["qps"] for g in ok] or [None])}
        optional.append("robustness")
    fo = result.get("one_process_fanout")
    if isinstance(fo, dict):
        h["one_process_fanout"] = ({"error": str(fo["error"])[:80]} if "error" in fo else
                                   {"segments": fo.get("segments"), "devices": len(fo.get("devices", [])),
                                    "build_points_per_s": fo.get("build", {}).get("points_per_s"),
                                    "search_qps_collection": fo.get("search", {}).get("qps_collection"),
                                    "merged_equals_host_merge": fo.get("search", {}).get("merged_equals_host_merge")})
    if isinstance(result.get("configs"), dict):
        h["configs"] = _configs_summary(result["configs"])
    if details_path:
        h["details"] = os.path.relpath(details_path, ROOT) if os.path.isabs(details_path) else details_path
    for key in ["robustness", "batch_sweep", "one_process_fanout", "configs", "checks"]:
        if len(json.dumps(h)) <= HEADLINE_MAX_BYTES:
            break
        if key in h:
            h[key] = "see details"
    return h


def emit(result, details_path):
    """details -> file (+ stderr), compact headline -> the LAST stdout line"""
    try:
        with open(details_path, "w") as f:
            json.dump(result, f)
            f.write("\n")
        scratch = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(scratch):
            with open(os.path.join(scratch, "bench_details.json"), "w") as f:
                json.dump(result, f)
                f.write("\n")
    except OSError as e:
        print("bench.py: could not write %s: %r" % (details_path, e), file=sys.stderr)
    print("bench_details " + json.dumps(result), file=sys.stderr, flush=True)
    line = json.dumps(headline(result, details_path))
    assert len(line) <= HEADLINE_MAX_BYTES, len(line)
    sys.stdout.flush()
    print(line, flush=True)


# ------------------------------------------------------------------------------------------------------------------------
def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    import torch
    import torch.distributed as dist
    import numpy as np

    hip = not args.test_backend
    how = resolve_world(args, argv, device_count=torch.cuda.device_count() if hip else None)
    if how[0] == "fail":
        print("bench.py: " + how[1], file=sys.stderr, flush=True)
        sys.exit(2)
    if how[0] == "exec":
        print("bench.py: --gpus %d without a launcher: starting %d ranks: %s" % (args.gpus, args.gpus, " ".join(how[1])), file=sys.stderr, flush=True)
        os.execv(how[1][0], how[1])
    _, rank, local_rank, world = how

    from qdrant_amd import sharded
    if hip:
        import qdrant_amd as qa
        from qdrant_amd import _ffi as F
        if args.ranks_share_gpu:
            local_rank = local_rank % max(1, torch.cuda.device_count())      # (from here on: the DEVICE this rank runs on)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    rccl_ranks = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if hip and not args.ranks_share_gpu:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        # prove the collective spans N ranks before anything is timed: a real all-gather of one word per rank
        probe = torch.zeros(world, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(probe, torch.full((1,), rank + 1, dtype=torch.int32, device=dev))
        if hip:
            torch.cuda.synchronize(dev)
        rccl_ranks = int((probe > 0).sum().item())
        assert probe.tolist() == list(range(1, world + 1)), probe.tolist()
    assert rccl_ranks == world == args.gpus, (rccl_ranks, world, args.gpus)

    dim, Q, top = args.dim, args.batch, args.top
    seed = 0x5EED0002  # 0x5EED0000 + config id
    strong = world > 1 and args.scaling == "strong"
    if strong:
        row0, n = sharded.row_split(args.rows)          # this rank's slice of the ONE segment
        row_seed = seed
    else:
        row0, n = 0, args.rows
        row_seed = seed + 16 * rank                     # this rank's own segment
    nbatches = max(1, args.nqueries // Q)

    if hip:
        lib = F.lib()
        # ---- the stored block: generated and normalised on device, adopted without copying ----
        rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
        F.check(lib.qmx_synth_fill_f32(local_rank, row_seed, row0, n, dim, F.ptr(rows)))
        F.check(lib.qmx_preprocess_f32(local_rank, int(qa.Distance.Cosine), F.ptr(rows), n, dim, F.ptr(rows)))
        copy_flag = {"none": 0, "pair": F.SEG_SPLIT_COPY, "half": F.SEG_HALF_COPY, "i8": F.SEG_I8_COPY, "auto": F.SEG_AUTO_COPY}[args.split_copy]
        storage = qa.VectorStorage(rows, qa.Distance.Cosine, device_id=local_rank, flags=copy_flag)
        seg_info = storage.info()       # which copy the segment holds (under "auto": the library's own choice, measured at create) and what the trial saw
        eff_flag = {"i8": F.SEG_I8_COPY, "half": F.SEG_HALF_COPY, "pair": F.SEG_SPLIT_COPY, None: 0}[seg_info["derived_copy"]]
        queries = torch.empty((nbatches * Q, dim), dtype=torch.float32, device=dev)
        F.check(lib.qmx_synth_fill_f32(local_rank, seed + 1, 0, nbatches * Q, dim, F.ptr(queries)))
        stream = torch.cuda.Stream(dev)  # every kernel, the RCCL gather and the merge are ordered on this stream
        torch.cuda.set_stream(stream)
        backend = sharded.HipBackend(storage, Q, local_rank, stream)      # owns the qmx_query of this rank
        qh = backend.qh
        F.check(lib.qmx_query_set_timing(qh, 1))
    else:
        # tests only: an injected compute backend (module:factory) drives the launcher, the collective, the merge contract and the line on CPU
        mod, fn = args.test_backend.split(":")
        backend, queries = getattr(importlib.import_module(mod), fn)(rank=rank, world=world, row_seed=row_seed, row0=row0, n=n, dim=dim,
                                                                     nqueries=nbatches * Q, query_seed=seed + 1)
        seg_info, eff_flag, copy_flag, stream = {}, 0, 0, None
    searcher = sharded.ShardedSearcher(backend, n, Q, top, device=dev)  # scan -> ONE all-gather -> merge (world > 1)
    out, counts = searcher.out, searcher.counts
    # Batches in flight: every lane is a query handle + its stream + its result buffers (+ at N > 1 its own ShardedSearcher: record, gathered records,
    # merged lists); consecutive steps alternate between the lanes - i % len(lanes), the same order on every rank, so the ranks issue their collectives in
    # the same order.  Every step is still one whole search of one batch - the GPU merely has the next batch's head to run while this batch's scans, its
    # all-gather and its merge are in flight.  N = 1 and N > 1 run the SAME experiment: the same lanes, + per step one collective and one merge.
    lanes = [(backend, searcher)]
    for _ in range(args.in_flight - 1):
        if hip:
            backend2 = sharded.HipBackend(storage, Q, local_rank, torch.cuda.Stream(dev))
            F.check(lib.qmx_query_set_timing(backend2.qh, 1))
        else:
            backend2 = backend        # the injected CPU backend is stateless between calls
        lanes.append((backend2, sharded.ShardedSearcher(backend2, n, Q, top, device=dev)))

    def step(i):
        b = i % nbatches
        qb = queries[b * Q:(b + 1) * Q]
        be, se = lanes[i % len(lanes)]
        if not hip:
            se.search(qb)
            return
        with torch.cuda.stream(be.stream):
            if world > 1:
                se.search(qb)
            else:
                be.local_topk(qb, top, se.out, se.counts)

    def fence():
        if hip:
            torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            if hip:
                torch.cuda.synchronize(dev)

    # set-up, untimed: a short steady-state run (the driver's form of the command has 5 warm-up steps = 7 ms of work after minutes of data generation: the first
    # timed steps then run on ramping clocks with one batch in flight: 83 k QPS at --steps 20 against 87 - 88 k at --steps 100 on the same library)
    prewarm_steps = 0
    if args.prewarm_ms > 0 and world > 1:
        # (every rank must issue the SAME number of steps - a step holds a collective -, so the count is derived from the flag, not timed:
        # --prewarm-ms / 1.5 ms per step, a whole number of rounds over the lanes)
        fixed = max(len(lanes), int(math.ceil(args.prewarm_ms / 1.5 / len(lanes))) * len(lanes))
        for prewarm_steps in range(1, fixed + 1):
            step(prewarm_steps - 1)
        fence()
    elif args.prewarm_ms > 0:
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms and prewarm_steps < 4096:
            step(prewarm_steps)
            prewarm_steps += 1
            if prewarm_steps % 16 == 0:
                fence()
        fence()
    for i in range(args.warmup):
        step(i)
    fence()
    ms0, l0 = C.c_float(), C.c_uint32()
    if hip:
        for be, _ in lanes:
            F.check(lib.qmx_query_timing(be.qh, C.byref(ms0), C.byref(l0)))  # drop warm-up launches

    fence()
    # (the spread of the timed region, without touching it: every `gsz` steps an event on EVERY lane's stream, read after the closing fence; a group ends
    # when its last lane does)
    # (a group is at least two whole rounds over the lanes: the batches in flight finish out of order by up to a step, a shorter group measures that jitter)
    gsz = len(lanes) * max(2, int(round(args.steps / 8.0 / len(lanes))))
    calls0 = sharded.COLLECTIVE_CALLS

    def mark():
        ev = []
        for be, _ in lanes:
            e = torch.cuda.Event(enable_timing=True)
            e.record(be.stream)
            ev.append(e)
        return ev
    marks = [mark()] if hip else []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        if hip and (i + 1) % gsz == 0:
            marks.append(mark())
    fence()
    elapsed = time.perf_counter() - t0
    collectives_per_step = (sharded.COLLECTIVE_CALLS - calls0) / float(max(1, args.steps))
    at = [max(marks[0][0].elapsed_time(e) for e in ev) for ev in marks]      # when each group's last lane passed its mark (ms after the first mark)
    group_ms = [(at[j + 1] - at[j]) / gsz for j in range(len(at) - 1) if at[j + 1] > at[j]]        # device time per step, per group of gsz steps

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # value = queries per second against the whole COLLECTION (at N > 1, weak: N segments of --rows rows each = one collection of N x rows; every query is
    # answered over all of it).  Ideal weak scaling reads value(N) = value(1): the collection grows with N at constant QPS, as the reference's
    # segment-parallel model implies (segments_searcher.rs:250-285).  (query, segment) searches per second = N x value stays in config.
    value = Q * args.steps / elapsed

    if world == 1:
        workload = "C2: 1 segment %s x d=%d f32 cosine, brute-force exact top-%d, batch Q=%d" % (_human(n), dim, top, Q)
        vecs = _human(args.rows)
    elif strong:
        workload = ("C2 row-split: ONE segment %s x d=%d f32 cosine split by contiguous row range over %d GPUs (%s rows each), top-%d, batch Q=%d, "
                    "one RCCL all-gather + merge per step" % (_human(args.rows), dim, world, _human(n), top, Q))
        vecs = _human(args.rows)
    else:
        workload = ("C5: %d segments (one per GPU) x %s x d=%d f32 cosine = one collection of %s vectors, top-%d, batch Q=%d, one RCCL all-gather + "
                    "merge per step" % (world, _human(n), dim, _human(world * n), top, Q))
        vecs = "%d x %s" % (world, _human(args.rows))
    result = {
        "metric": "QPS @ recall@10, brute-force, d=%d %s vecs, f32 cosine top-%d" % (dim, vecs, top),
        "value": round(value, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        # spread over groups of steps inside the timed region (device time between stream events): standard deviation of the groups' QPS
        # (one batch in flight only: with several, the lanes' streams run ahead of and behind one another by whole steps - the GPU takes their kernels in its own
        # order - so the time between two rounds of marks measures that order, not the throughput; measured: "46 k QPS" of spread on a steady 92 k)
        "value_stddev": round(_stddev([Q / (m * 1e-3) for m in group_ms if m > 0]), 2) if (len(group_ms) >= 3 and len(lanes) == 1) else None,
        "step_groups": ({"steps_per_group": gsz, "ms_per_step_min": round(min(group_ms), 4) if group_ms else None,
                         "ms_per_step_max": round(max(group_ms), 4) if group_ms else None} if len(lanes) == 1 else None),
        "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "rccl_ranks": rccl_ranks,
        "collective": ("none (one GPU)" if world == 1 else "gloo over device tensors (--ranks-share-gpu: a test of the path, not RCCL)" if args.ranks_share_gpu else
                       "RCCL over xGMI (torch.distributed nccl)" if hip else args.backend),
        "config": {"workload": workload,
                   "rows_per_gpu": n, "dim": dim, "batch": Q, "top": top, "distinct_queries": nbatches * Q,
                   "unit_of_value": ("queries per second against the ONE row-split segment" if strong else
                                     "queries per second against the whole collection (N segments of rows_per_gpu rows; at n_gpus=1 plain QPS on C2); "
                                     "ideal weak scaling: value(N) == value(1)"),
                   "collection_qps": round(value, 2),
                   "segment_searches_per_s": round(value * (1 if strong else world), 2),
                   "batches_in_flight": len(lanes), "prewarm_steps": prewarm_steps,
                   "collectives_per_step": round(collectives_per_step, 3)},
    }
    if args.ranks_share_gpu:
        result["data"] = "synthetic; %d ranks SHARE %d device(s): a test of the N > 1 path, not a scaling measurement" % (world, torch.cuda.device_count())
    if not hip:
        # the launcher / collective / line under test: whatever the injected backend computed is NOT a measurement
        result["data"] = "TEST BACKEND %s on CPU: not a measurement" % args.test_backend
        result["dtype"] = "test"
        last = lanes[(args.steps - 1) % len(lanes)][1]           # the lane of the last step
        result["merged_checksum"] = int(last.merged.to(torch.int64).sum().item())
        result["lanes_used"] = sorted(set(i % len(lanes) for i in range(args.steps)))
        if rank == 0:
            emit(result, args.details)
        if world > 1:
            dist.destroy_process_group()
        return

    from bench_sections import (c1_section, c3_section, c4_section, cpu_baseline, iid_walk_leg, derived_copy_point, hbm_point, one_process_fanout, robustness, tq_section,
                                _counters_dict)
    kms, kl = C.c_float(), C.c_uint32()
    for be, _ in lanes:       # the scan launches of every batch in flight
        m1, l1 = C.c_float(), C.c_uint32()
        F.check(lib.qmx_query_timing(be.qh, C.byref(m1), C.byref(l1)))
        kms.value += m1.value
        kl.value += l1.value
    kernel_symbol = F.last_kernel(qh)
    kernel_ms = kms.value / max(1, kl.value)
    # With several batches in flight an event pair around a scan also holds the time the launch WAITED behind another batch's scan (one block per CU: the scans
    # take turns), so the timed region's figure is not the kernel's duration.  The kernel's own: the same local searches with ONE batch in flight, right here
    # (untimed; N > 1: every rank on its own device at once, no collective inside; the slowest rank's figure is reported).
    kernel_ms_in_flight = kernel_ms
    if len(lanes) > 1:
        be0, se0 = lanes[0]
        fence()
        F.check(lib.qmx_query_timing(be0.qh, C.byref(C.c_float()), C.byref(C.c_uint32())))      # reset
        for i in range(16):
            b = i % nbatches
            with torch.cuda.stream(be0.stream):
                be0.local_topk(queries[b * Q:(b + 1) * Q], top, se0.out, se0.counts)
        torch.cuda.synchronize(dev)
        m1, l1 = C.c_float(), C.c_uint32()
        F.check(lib.qmx_query_timing(be0.qh, C.byref(m1), C.byref(l1)))
        if l1.value:
            kernel_ms = m1.value / l1.value
    if world > 1:
        km = torch.tensor([kernel_ms, kernel_ms_in_flight], dtype=torch.float64, device=dev)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kernel_ms, kernel_ms_in_flight = float(km[0].item()), float(km[1].item())
    # bytes the dominant kernel of the timed step has to read per launch: the f32 block (3072 B/row at d=768) for the exact scans; the derived
    # copy the prefilter scans (QMX_SEG_I8_COPY: 1 B / element, QMX_SEG_HALF_COPY: 2 B, QMX_SEG_SPLIT_COPY: 4 B) when that is the kernel that ran
    half_copy = "scan_f16pair_kernel<true>" in kernel_symbol or "scan_f16half256_kernel" in kernel_symbol
    tile_q = 256.0 if "scan_f16half256_kernel" in kernel_symbol else 128.0            # queries per pass of the prefilter shape that ran
    i8_copy = "scan_i8copy_kernel" in kernel_symbol
    elem_bytes = 1 if i8_copy else 2 if half_copy else 4
    row_bytes = dim * elem_bytes
    launches_per_step = max(1, int(kl.value)) / float(max(1, args.steps))
    # the prefilter over a derived copy may cover the block in more than one launch of the same kernel: bytes and flops per launch are the per-launch
    # AVERAGES, like kernel_ms, so that achieved = sum of bytes / sum of kernel time
    prefilter = "scan_f16pair_kernel" in kernel_symbol or "scan_f16half256_kernel" in kernel_symbol or i8_copy
    launches_per_pass = max(1.0, launches_per_step / math.ceil(Q / tile_q)) if prefilter else 1.0
    alg_bytes = int(n * row_bytes / launches_per_pass)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    result["config"]["timed_path"] = _timed_path(kernel_symbol)
    result["config"]["derived_copy"] = dict(seg_info, requested=args.split_copy)
    if prefilter:
        result["dtype"] = "f32 (%s prefilter + exact f32 re-score)" % ("int8" if i8_copy else "f16")
    result["roofline"] = _roofline(n, dim, kernel_ms, alg_bytes, achieved, int(kl.value), launches_per_step, Q, kernel_symbol, launches_per_pass)
    result["roofline"]["kernel_ms_of"] = ("one batch in flight (16 steps right after the timed region): the kernel's own duration" if kernel_ms != kernel_ms_in_flight
                                          else "the timed region")
    result["roofline"]["kernel_ms_in_timed_region"] = round(kernel_ms_in_flight, 4)     # (event pairs there include the wait behind the other batches' scans)

    if world > 1:
        # the three stages of a step, measured in their own untimed run (stream events around local search / all-gather / merge on every lane)
        for _, se in lanes:
            se.timing = True
        for i in range(16 * len(lanes)):
            step(i)
        fence()
        per = [se.stage_us() for _, se in lanes]
        mean = lambda k: sum(p[k] for p in per) / len(per)      # noqa: E731
        mine = torch.tensor([mean("local_search_us"), mean("allgather_us"), mean("merge_us")], dtype=torch.float64, device=dev)
        worst = mine.clone()
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        for _, se in lanes:
            se.timing = False
        result["config"]["per_step_us"] = {"of": "rank 0, mean over %d searches per lane, stream events; max over ranks beside it" % per[0]["searches"],
                                           "local_search_us": round(float(mine[0]), 2), "allgather_us": round(float(mine[1]), 2), "merge_us": round(float(mine[2]), 2),
                                           "allgather_us_max_over_ranks": round(float(worst[1]), 2), "local_search_us_max_over_ranks": round(float(worst[0]), 2)}
        # Everything below is rank 0's own side measurement (the block-stream roofline of its segment, the one-process fan-out over every visible
        # device): the process group ends HERE, so no rank sits in an RCCL barrier - a kernel spinning on a device - while rank 0 drives that device.
        fence()
        dist.destroy_process_group()
        if rank != 0:
            for be, _ in lanes:
                be.close()
            return
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
    if rank == 0 and not args.no_hbm_point:
        # SURVEY 8(d) / north_star (>= 70 % of the HBM roofline on C2): the EXACT scan streaming the stored f32 block itself, once for 16 queries and
        # once for 1; outside the timed region, same rows, same measurement (HIP events on the kernel's stream).  (N > 1: rank 0 measures its segment
        # after the process group has ended; the other ranks have left.)
        qa.set_option("no_split_scan", 1)
        try:
            for Qh, key in ((BLOCK_STREAM_BATCH, "roofline_hbm_point_q16"), (1, "roofline_hbm_point_q1")):
                try:
                    result[key] = hbm_point(Qh, storage, queries, n, dim, top, local_rank, stream, lib, F, sharded, torch, steps=30)
                except Exception as e:
                    result[key] = {"error": repr(e)[:300]}
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
        # the same search over the OTHER derived copy of the block - the f16 half copy (2 B / element, band 1e-3 |q| |row|) when the int8 copy (1 B /
        # element, worst-case band of the two roundings) is the timed one, and vice versa: lists checked against the exact scan inside the leg
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
        # and the sharded search with its merge, over one segment per visible device of this run (world > 1: after the process group has ended and
        # the other ranks have left their devices; a single GPU: two segments on it, which exercises the same code path)
        try:
            result["one_process_fanout"] = one_process_fanout(args, 1 if args.ranks_share_gpu else world, dim, Q, top, lib, F, qa, torch, np)
        except Exception as e:
            result["one_process_fanout"] = {"error": repr(e)[:400]}
    if solo and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(args, rows, queries, out, counts, n, dim, Q, top, lib, qh, F, qa, np, torch)
    for be, _ in lanes:
        be.close()
    wanted = [c for c in args.configs.lower().split(",") if c]
    if solo and wanted:
        cfg = {}
        lanes.clear()                   # the lanes hold the backends, the backends the segment: release the block's copies before the other configs
        be = se = _ = backend2 = None   # noqa: F841  (loop variables above)
        del searcher, backend, storage, out, counts
        ctx = dict(args=args, dev=dev, lib=lib, F=F, qa=qa, np=np, torch=torch)
        if "c1" in wanted:
            try:
                cfg["C1"] = c1_section(ctx)
            except Exception as e:
                cfg["C1"] = {"error": repr(e)[:400]}
        if "c3" in wanted and args.iid_walk:
            try:        # (before C3 refills the block: the C3 walk on the rows SURVEY 8(d) names - C2's own iid rows)
                cfg["C3_walk_on_iid_rows"] = iid_walk_leg(ctx, rows, queries)
            except Exception as e:
                cfg["C3_walk_on_iid_rows"] = {"error": repr(e)[:400]}
            torch.cuda.empty_cache()
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
    emit(result, args.details)      # (rank 0: the other ranks returned when the process group ended)


if __name__ == "__main__":
    main()
