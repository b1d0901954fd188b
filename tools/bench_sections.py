"""The legs of bench.py outside the timed region (rank 0, N = 1): the C2 secondary points (exact-stream points, the other derived copy, batch sweep,
robustness families, one-process fan-out, CPU baseline) and BASELINE.json's other single-GPU configs (C3, TQ4, C4).  Their results go to
bench_details.json; bench.py's last stdout line carries a few numbers of each (bench.headline)."""
import ctypes as C
import os
import sys
import time

from bench_roofline import (HBM_PEAK_GBPS, MFMA_F16_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS, MFMA_I8_PEAK_TOPS, ROOT, _attach_traffic, _human)  # noqa: F401

QUERY_ROW0 = 1 << 40   # latent-model queries: rows of the same generator (same basis), far past the stored range


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
            "sample_short": "oracle peek_top_iter, first %s of %s rows, Q=%d, scaled" % (_human(S), _human(n), Q),
            "kind_note": "the oracle's C restatement of the reference's AVX2+FMA scorer and peek_top_iter loop, not the Rust binary (no cargo in the image)",
            "sample": "oracle peek_top_iter (AVX2+FMA dot, 64-id chunks, heap of %d) over the first %d of %d rows, Q=%d, "
                      "%d threads on disjoint row ranges (usable cores: affinity + cgroup quota; os.cpu_count() = %d), %d scans in %.1f s; "
                      "QPS scaled by %d/%d to the full segment" % (top, S, n, Q, cores, os.cpu_count() or 0, reps, el, S, n),
            "gflops_all_cores": round(flops * S * Q * reps / el / 1e9, 1),
            "single_thread": {"value": round(qps1, 4), "unit": "queries/s", "cores": 1,
                              "gflops": round(flops * S1 * Q * reps1 / el1 / 1e9, 2), "ns_per_row_per_query": round(el1 / (reps1 * S1 * Q) * 1e9, 2),
                              "sample": "the same loop, one thread, first %d rows, %d scans in %.1f s, scaled by %d/%d" % (S1, reps1, el1, S1, n)},
            "gpu_matches_oracle_on_sample_bit_exact": ok}



def c1_section(ctx, seconds=4.0):
    """BASELINE.json configs[0] ("C1"): one segment of 100 000 x 128 f32, cosine, brute-force exact top-10 on the CPU RawScorer - the reference's own
    CPU-runnable case (lib/segment/benches/vector_search.rs:21,34-104).  Here: the oracle's restatement of that path (AVX2+FMA CosineMetric +
    peek_top_iter, kind "port") timed on this box's host cores at 1 thread and at all usable cores, 1 024 distinct queries in batches of 32;
    and the device through the C-ABI at Q in {1, 8, 32} on the same rows, its lists checked against the oracle's bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    args, lib, F, qa, np, torch = (ctx[k] for k in ("args", "lib", "F", "qa", "np", "torch"))
    n, dim, nq, top, seed = 100_000, 128, 1024, 10, 0x5EED0001
    rows = O.preprocess(O.COSINE, O.synth(seed, 0, n, dim))
    queries = O.synth(seed + 1, 0, nq, dim)
    ost = O.DenseStorage(O.F32, O.COSINE, rows)
    enc = ost.encode_queries(queries)
    cores = usable_cores()
    cpu = {}
    for name, threads, budget in (("one_thread", 0, seconds * 0.5), ("all_cores", cores, seconds * 0.5)):
        done, t0 = 0, time.perf_counter()
        while True:
            b = (done // 32) % (nq // 32)
            ost.peek_top(enc[b * 32:(b + 1) * 32], top, encoded=True, threads=threads)
            done += 32
            el = time.perf_counter() - t0
            if el >= budget:
                break
        cpu[name] = {"qps": round(done / el, 1), "threads": max(1, threads), "queries": done, "seconds": round(el, 2),
                     "ns_per_row_per_query": round(el / (done * n) * 1e9, 3), "GBps": round(done * n * dim * 4 / 32 / el / 1e9, 2)}
    want = ost.peek_top(enc, top, encoded=True, threads=cores)
    st = qa.VectorStorage(rows, qa.Distance.Cosine, device_id=ctx.get("device_id", 0))
    dev = {}
    ok = True
    for Q in (1, 8, 32):
        count = nq if Q == 32 else 256 if Q == 8 else 64
        searchers = [qa.BatchFilteredSearcher(queries[q0:q0 + Q], st, top) for q0 in range(0, count, Q)]
        got = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in searchers:
            got += s.peek_top_all()
        el = time.perf_counter() - t0
        for j, g in enumerate(got):
            ok = ok and g["idx"].tolist() == want[j]["idx"].tolist() and np.array_equal(g["score"].view(np.uint32), want[j]["score"].view(np.uint32))
        dev["Q%d" % Q] = {"qps_wall_sync_per_batch": round(count / el, 1), "queries": count, "kernel": F.last_kernel(searchers[-1].scorer._h)}
    if not ok:
        print("PARITY FAILURE: C1 device lists differ from the oracle's", file=sys.stderr)
    return {"workload": "C1: 1 segment 100k x d=128 f32 cosine, brute-force exact top-10, 1024 queries (rows iid N(0,1), normalised; seed 0x5EED0001)",
            "cpu_oracle_port": dict(cpu, cores=cores, kind="port",
                                    note="BASELINE's C1 is this CPU path: the oracle's C restatement of the AVX2+FMA scorer + peek_top_iter, batches of 32"),
            "device": dev, "device_equals_oracle_bit_exact": bool(ok),
            "device_note": "every search is a blocking call on host buffers (query upload + scan of 51.2 MB + list download): latency-bound, not a roofline point"}


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


def _reference_order_cost(ctx, scorer, raw, top, graph, ef, default_stats):
    """The same walk (no rescoring: the walk's own lists) with option hnsw_reference_heap_order: what "give me the reference's lists among equal scores"
    costs beside the default walk (SearchContext's two binary heaps kept on the device, search_context.rs:8-40)."""
    qa = ctx["qa"]
    qa.set_option("hnsw_reference_heap_order", 1)
    try:
        _, st, scored = _timed_quantized(ctx, scorer, raw, top, 0.0, False, graph, ef, 2, default_stats["roofline"]["bytes_per_scored_row"])
    except Exception as e:
        return {"error": repr(e)[:200]}
    finally:
        qa.set_option("hnsw_reference_heap_order", -1)
    return {"kernel": st["kernel"], "kernel_ms": st["kernel_ms"], "qps_wall": st["qps_wall"], "frac": st["roofline"]["frac"],
            "over_default_walk": round(st["kernel_ms"] / default_stats["kernel_ms"], 2) if default_stats.get("kernel_ms") else None,
            "searches_per_launch": scorer.nq, "points_scored_per_query": round(scored / scorer.nq, 1)}


def _oracle_walk_check(qa, np, graph, scorer, walker, oracle_walk, nchk, top, ef):
    """The CPU oracle walks THE SAME graph with its scorer (host copy of the codes + links).  Three comparisons over `nchk` searches:
      default walk      same_ids / same_score_bits of the first `top` results (the device orders equal scores by ascending id: lists may differ at ties);
      tie_explained     per search whose pop sequence differs from the oracle's: the first differing position holds two bit-equal scores
                        (tests/parity_asserts.first_divergence_is_a_tie) AND the reference-order run of that search is the oracle's walk - anything
                        else is printed as a PARITY FAILURE and listed;
      reference order   option hnsw_reference_heap_order (the reference's two binary heaps on the device): lists AND pop sequences must be the oracle's.
    Walks run with top = ef so that the returned list is the whole `nearest` (its last score is the bound when a walk ends)."""
    from parity_asserts import first_divergence_is_a_tie
    bits = lambda x: np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)      # noqa: E731
    t0 = time.perf_counter()
    walker.pops = []
    want = oracle_walk(ef, ef)
    want_pops, walker.pops = walker.pops, None
    got, got_pops = graph.search_traced(ef, ef, scorer)
    same_ids = sum(int(a["idx"][:top].tolist() == b["idx"][:top].tolist()) for a, b in zip(got, want))
    same_bits = sum(int(np.array_equal(bits(a["score"][:top]), bits(b["score"][:top]))) for a, b in zip(got, want))
    # the same searches with the reference's heaps on the device first: a search whose default walk parts from the oracle's at a tie is "explained" ONLY IF its
    # reference-order run then IS the oracle's walk, pops and lists (a defect behind an early tie would show there: the first divergence alone proves little)
    qa.set_option("hnsw_reference_heap_order", 1)
    try:
        ref, ref_pops = graph.search_traced(ef, ef, scorer)
    finally:
        qa.set_option("hnsw_reference_heap_order", -1)
    ref_same = [bool(a["idx"].tolist() == b["idx"].tolist() and np.array_equal(bits(a["score"]), bits(b["score"]))
                     and ap["idx"].tolist() == bp["idx"].tolist() and np.array_equal(bits(ap["score"]), bits(bp["score"])))
                for a, b, ap, bp in zip(ref, want, ref_pops, want_pops)]
    differing, explained, bad = 0, 0, []
    for qi in range(nchk):
        v = first_divergence_is_a_tie(got_pops[qi], want_pops[qi], bound_score=got[qi]["score"][-1] if len(got[qi]) == ef else None)
        if v != "same":
            differing += 1
            if v == "tie" and not ref_same[qi]:
                v = "a tie at the first divergence, but the reference-order run of the same search is not the oracle's walk"
            explained += v == "tie"
            if v != "tie":
                bad.append("search %d: %s" % (qi, v))
    out = {"same_ids": "%d/%d" % (same_ids, nchk), "same_score_bits": "%d/%d" % (same_bits, nchk),
           "pop_sequences_equal": "%d/%d" % (nchk - differing, nchk), "tie_explained": "%d/%d" % (explained, differing)}
    if bad:
        out["unexplained"] = bad[:8]
        print("PARITY FAILURE: HNSW walk differs from the oracle away from a tie: %s" % bad[:3], file=sys.stderr)
    out["reference_heap_order_same_ids"] = "%d/%d" % (sum(int(a["idx"].tolist() == b["idx"].tolist()) for a, b in zip(ref, want)), nchk)
    out["reference_heap_order_same_score_bits"] = "%d/%d" % (sum(int(np.array_equal(bits(a["score"]), bits(b["score"]))) for a, b in zip(ref, want)), nchk)
    out["reference_heap_order_same_pops"] = "%d/%d" % (sum(int(a["idx"].tolist() == b["idx"].tolist() and np.array_equal(bits(a["score"]), bits(b["score"])))
                                                           for a, b in zip(ref_pops, want_pops)), nchk)
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def iid_walk_leg(ctx, rows, queries):
    """SURVEY 8(d) names C3's rows as C2's: iid N(0,1) coordinates, normalised.  HNSW recall on such rows is hopeless (no neighbourhood structure in 768
    dimensions), which is why the C3 / C4 legs run on rows of low intrinsic dimension - but walk THROUGHPUT depends on the data (hops per search, what the beam
    admits), so this leg runs the C3 walk once on the rows the contract names, for the record: SQ int8 codes of the C2 block, graph built through the SQ
    scorer (m = 16, ef_construct = 100), ef = 128, no rescoring."""
    args, dev, lib, F, qa, np, torch = (ctx[k] for k in ("args", "dev", "lib", "F", "qa", "np", "torch"))
    n, dim = rows.shape
    top, nq_h = 10, min(args.hnsw_queries, int(queries.shape[0]))
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    quant = qa.ScalarQuantizer.fit(rows, dim, qa.Distance.Dot)
    p = quant.params()
    codes = torch.empty((n, quant.quantized_vector_size()), dtype=torch.uint8, device=dev)
    F.check(lib.qmx_sq_encode(dev.index or 0, int(qa.Distance.Dot), C.byref(p), F.ptr(rows), n, dim, F.ptr(codes)))
    enc = qa.EncodedVectorsU8(codes, quant)
    del codes
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    graph = qa.GraphLayers.build(enc, m=16, ef_construct=100, seed=42)
    t_build = time.perf_counter() - t0
    qh = queries[:nq_h].contiguous()
    scorer, raw = qa.new_raw_scorer(qh, enc), qa.new_raw_scorer(qh, vs)
    res, st, scored = _timed_quantized(ctx, scorer, raw, top, 0.0, False, graph, 128, 3, quant.quantized_vector_size())
    n_gt = min(256, nq_h)
    exact = qa.BatchFilteredSearcher(queries[:n_gt].cpu().numpy(), vs, top).peek_top_all()
    st.update({"rows": "the C2 block: %s x d=%d iid N(0,1), normalised (SURVEY 8d's C3 rows)" % (_human(n), dim), "m": 16, "ef_construct": 100, "ef": 128,
               "searches_per_launch": nq_h, "build_s": round(t_build, 2), "points_scored_per_query": round(scored / nq_h, 1),
               "recall_at_10_vs_exact": round(_recall(res[:n_gt], exact, top), 4),
               "note": "recall on iid 768-d rows is what HNSW gives there at any ef a search can afford; the figure that matters is kernel_ms / frac beside the latent-row leg's"})
    return st


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
    for Qb in (1, 32, 128):
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
    if "Q32" in bf and "Q128" in bf and bf["Q32"].get("kernel_ms") and bf["Q128"].get("kernel_ms"):
        bf["Q128"]["queries_per_ms_of_scan_vs_Q32"] = round((128 / bf["Q128"]["kernel_ms"]) / (32 / bf["Q32"]["kernel_ms"]), 2)
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
    st["reference_heap_order"] = _reference_order_cost(ctx, scorer, raw, top, graph, 128, st)
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
            nchk = min(256, int(queries.shape[0]))
            qchk = queries[:nchk].cpu().numpy()
            st["oracle_walk_check"] = _oracle_walk_check(qa, np, graph, qa.new_raw_scorer(queries[:nchk].contiguous(), enc), walker,
                                                         lambda t, e: walker.search_sq(flags, osq_all, qchk, t, e), nchk, 2 * top, 128)
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
    for Qb in (1, 32, 128):
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
        if Qb == 128 and stats.get("kernel_ms"):
            # the 128-query pass (scan_tq4w.hip) is bound by the matrix cores, not by its 3.9 GB of codes: 2 digits x 2 x 128 x d integer operations per row
            # against the dense int8 peak (MI355X guide: 5 033 TOP/s at 2.4 GHz; the chip holds ~1.93 GHz under this load)
            tops = 2.0 * 2 * n * dim * Qb / (stats["kernel_ms"] * 1e-3) / 1e12
            stats["matrix_cores"] = {"bound": "mfma", "achieved": round(tops, 1), "peak": 5033.0, "unit": "TOP/s (int8)", "frac": round(tops / 5033.0, 4)}
        bf["Q%d" % Qb] = stats
    if "Q32" in bf and "Q128" in bf and bf["Q32"].get("kernel_ms") and bf["Q128"].get("kernel_ms"):
        bf["Q128"]["queries_per_ms_of_scan_vs_Q32"] = round((128 / bf["Q128"]["kernel_ms"]) / (32 / bf["Q32"]["kernel_ms"]), 2)
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
    # ---- LUT build on the matrix cores (encode_query of a batch, qmx_query_update): device time of the update between two events on the query's stream
    # (cosine preprocess + pq_lut_mfma_lds_kernel; the host's share of the call is in ms_wall).  The kernel WRITES nq x m x 256 x 4 bytes of LUTs and does
    # 2 x 256 x d flop per query: priced against both roofs, it is bound by the write
    qh_all = queries[:nq_h].contiguous()
    scorer = qa.new_raw_scorer(qh_all, enc)
    lut_stream = torch.cuda.Stream(dev)
    F.check(lib.qmx_query_set_stream(scorer._h, C.c_void_p(lut_stream.cuda_stream)))
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    F.check(lib.qmx_query_update(scorer._h, F.ptr(qh_all)))
    F.check(lib.qmx_query_synchronize(scorer._h))
    t0 = time.perf_counter()
    e0.record(lut_stream)
    for _ in range(5):
        F.check(lib.qmx_query_update(scorer._h, F.ptr(qh_all)))
    e1.record(lut_stream)
    F.check(lib.qmx_query_synchronize(scorer._h))
    lut_wall = (time.perf_counter() - t0) / 5 * 1e3
    lut_ms = e0.elapsed_time(e1) / 5
    lut_flops = 2.0 * 256 * dim * nq_h                       # 2 x 256 x d flop per query
    lut_bytes = float(nq_h) * quant.m * 256 * 4 + float(nq_h) * dim * 4 * 3      # the LUTs written + the batch read, normalised, read again
    wr = lut_bytes / (lut_ms * 1e-3) / 1e9
    out["lut_build_mfma"] = {"queries": nq_h, "ms_device_incl_preprocess": round(lut_ms, 4), "ms_wall": round(lut_wall, 3), "flops": lut_flops, "bytes": lut_bytes,
                             "kernel_roofline": {"bound": "hbm (write)", "achieved": round(wr, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(wr / HBM_PEAK_GBPS, 4),
                                                 "kernel_ms": round(lut_ms, 4),
                                                 "mfma": {"achieved_TFLOPs": round(lut_flops / (lut_ms * 1e-3) / 1e12, 2), "peak_TFLOPs": MFMA_F32_PEAK_TFLOPS,
                                                          "frac": round(lut_flops / (lut_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)},
                                                 "note": "the contraction is 6.4 GFLOP for %d MB of LUTs written: the matrix cores make the table, the write bounds it" % (nq_h * 96 // 1024)}}
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
    hn["reference_heap_order"] = _reference_order_cost(ctx, scorer, raw, top, graph, 128, walks["no_rescoring"])
    try:
        # what the hop prefilter lets through (qmx_counters of qmx_hnsw_search: level-0 candidates that met the 8-bit bound / those scored exactly), and the hops
        # of a search (its pop sequence): the walk's exact scores cost one 4-byte gather per chunk and survivor - requests, not bytes, are what it is bound by
        graph.search(top, 128, scorer)
        offered, exact_scored = int(graph.counters.prefilter_candidates), int(graph.counters.verified_rows)
        _, pops = graph.search_traced(top, 128, qa.new_raw_scorer(queries[:256].contiguous(), enc))
        hops = float(np.mean([len(p_) for p_ in pops]))
        kms = walks["no_rescoring"]["kernel_ms"]
        reqs = nq_h * (exact_scored / float(nq_h) * quant.m + offered / float(nq_h) * 2 + hops * 3)      # LUT gathers + code rows (96 B: two sectors) + link rows
        hn["hop_prefilter"] = {"hops_per_search": round(hops, 1), "candidates_offered_per_hop": round(offered / float(nq_h) / hops, 2),
                               "survivors_per_hop": round(exact_scored / float(nq_h) / hops, 2), "survival": round(exact_scored / float(max(1, offered)), 4),
                               "lut_gathers_per_search": int(exact_scored / float(nq_h) * quant.m),
                               "memory_requests_per_launch_estimate": int(reqs),
                               "G_requests_per_s": round(reqs / (kms * 1e-3) / 1e9, 2) if kms else None,
                               "note": "requests = 4-byte LUT gathers of the survivors (m per survivor, each its own 64-byte sector) + two sectors per offered code row + the link "
                                       "rows; the roof to read it against is tools/micro/gather_roof's `lut4 chain` line (profiles/r6_gather_roof_lut4.txt), not 8 TB/s"}
    except Exception as e:
        hn["hop_prefilter"] = {"error": repr(e)[:200]}
    hn["recall_f32_walk_vs_ef"] = {str(ef): round(_recall(graph.search(top, ef, qa.new_raw_scorer(queries[:n_gt].contiguous(), vs)), exact, top), 4)
                                   for ef in (64, 128, 256)}
    # brute force over the codes for reference (what the quantizer alone can do on these rows)
    bfs = qa.new_raw_scorer(queries[:32].contiguous(), enc)
    bfr = qa.new_raw_scorer(queries[:32].contiguous(), vs)
    res, st, _ = _timed_quantized(ctx, bfs, bfr, top, 2.0, True, None, 0, 3, quant.m, n_rows_scanned=n)
    st["recall_at_10_vs_exact"] = round(_recall(res, exact[:32], top), 4)
    if "pq_prefilter_kernel" in st["kernel"] and st["kernel_ms"] > 0:
        # what bounds pq_prefilter_kernel is not the 0.96 GB it streams: one ds_read_b32 per (row, chunk, 4 queries) + two vector instructions per gather + one
        # int8 matrix instruction per 4 gathers.  The LDS serves a wave's ds_read_b32 in two groups of 32 lanes (MI355X guide): 32 lane-reads per clock and CU.
        # PMC of this kernel (profiles/r3_c4_pq_prefilter_q32_pmc.txt, unchanged since): no bank conflicts; LDS index unit active 41 %, vector ALU 49 %, matrix
        # cores 41 % of the kernel's cycles - an issue-bound kernel whose dependent ds_read -> mfma chains keep three pipes each about half busy
        m_pad = (quant.m + 31) // 32 * 32
        lane_reads = float(n) * (32 // 4) * m_pad
        cus, clk = torch.cuda.get_device_properties(dev).multi_processor_count, 2.4e9
        peak = cus * 32 * clk
        st["roofline"]["hbm_frac"] = st["roofline"]["frac"]
        st["roofline"]["lds"] = {"lane_reads_per_s": round(lane_reads / (st["kernel_ms"] * 1e-3), 0), "peak_lane_reads_per_s": peak,
                                 "frac": round(lane_reads / (st["kernel_ms"] * 1e-3) / peak, 4),
                                 "what": "ds_read_b32 lane-reads (one per row, chunk and 4 queries) against CUs x 32 lanes x 2.4 GHz",
                                 "pmc": "profiles/r3_c4_pq_prefilter_q32_pmc.txt: SQ_LDS_BANK_CONFLICT 0, SQ_LDS_IDX_ACTIVE 2.40e8, SQ_INSTS_LDS 1.2e8, SQ_INSTS_VALU 2.86e8, SQ_INSTS_MFMA 3.0e7 per launch"}
        st["roofline"]["bound"] = "lds"
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
            chk = _oracle_walk_check(qa, np, graph, qa.new_raw_scorer(queries[:nchk].contiguous(), enc_exact), walker,
                                     lambda t, e: walker.search_pq(flags, opq, qpre, t, e), nchk, 2 * top, 128)
            want = walker.search_pq(flags, opq, qpre, 2 * top, 128)
            got_mfma = graph.search(2 * top, 128, qa.new_raw_scorer(queries[:nchk].contiguous(), enc))
            chk.pop("seconds")
            hn["oracle_walk_check"] = dict({"codes_byte_exact_first_1000": enc_ok, "exact_lut_same_ids": chk.pop("same_ids"),
                                            "exact_lut_same_score_bits": chk.pop("same_score_bits")}, **chk,
                                           mfma_lut_same_id_sets="%d/%d" % (sum(int(set(a["idx"].tolist()) == set(b["idx"].tolist())) for a, b in zip(got_mfma, want)), nchk),
                                           mfma_lut_max_rel_score_err=float(max(np.max(np.abs(a["score"][:min(len(a), len(b))] - b["score"][:min(len(a), len(b))]) /
                                                                                       np.maximum(np.abs(b["score"][:min(len(a), len(b))]), 1e-30)) for a, b in zip(got_mfma, want))),
                                           seconds=round(time.perf_counter() - t0, 1))
        except Exception as e:
            hn["oracle_walk_check"] = {"error": repr(e)[:300]}
    out["hnsw_pq_walk"] = hn
    return out
