set -u
mkdir -p gpurun_out
# PQ walk at the C4 size (10 M x 1536, m = 96): the one-wave kernel uncapped / capped per CU, and the block-per-search kernel
cat > /tmp/pqwalk10m.py <<'PY'
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import qdrant_amd as qa
from qdrant_amd import _ffi as F
lib = F.lib(); dev = torch.device("cuda", 0)
n, dim, nq, top, ef = int(sys.argv[1]), 1536, 8192, 20, 128
gen = torch.Generator(device=dev); gen.manual_seed(0x5EED0003)
centres = torch.randn((4096, dim), generator=gen, device=dev); centres = centres / centres.norm(dim=1, keepdim=True)
def make(count):
    out = torch.empty((count, dim), dtype=torch.float32, device=dev)
    for s in range(0, count, 1 << 20):
        e = min(count, s + (1 << 20))
        x = torch.randn((e - s, dim), generator=gen, device=dev)
        idx = torch.randint(0, 4096, (e - s,), generator=gen, device=dev)
        out[s:e] = centres[idx] + x * (0.35 / dim ** 0.5)
    F.check(lib.qmx_preprocess_f32(0, int(qa.Distance.Cosine), F.ptr(out), count, dim, F.ptr(out)))
    return out
rows = make(n); queries = make(nq).cpu().numpy()
vs = qa.VectorStorage(rows, qa.Distance.Cosine)
t0 = time.time(); graph = qa.GraphLayers.build(vs, m=16, ef_construct=100, seed=42); tb = time.time() - t0
cen, _ = qa.pq_train(rows[:10000].cpu().numpy(), dim, 16, 256, max_iterations=100, accuracy=1e-5)
quant = qa.ProductQuantizer(dim, qa.Distance.Dot, 16, cen); p = quant.params()
codes = torch.empty((n, quant.m), dtype=torch.uint8, device=dev)
F.check(lib.qmx_pq_encode(0, C.byref(p), F.ptr(rows), n, dim, F.ptr(codes)))
enc = qa.EncodedVectorsPQ(codes, quant)
scorer = qa.new_raw_scorer(queries, enc)
F.check(lib.qmx_query_set_timing(scorer._h, 1))
ref = None
print(json.dumps({"rows": n, "build_s": round(tb, 1)}), flush=True)
for name, opts in [("one_wave_uncapped", {}), ("one_wave_cap8", {"hnsw_pq_per_cu": 8}), ("one_wave_cap6", {"hnsw_pq_per_cu": 6}), ("one_wave_cap4", {"hnsw_pq_per_cu": 4}),
                   ("one_wave_cap12", {"hnsw_pq_per_cu": 12}), ("block_w8", {"no_hnsw_pq_block": 0}), ("block_w4", {"no_hnsw_pq_block": 0, "hnsw_pq_block_waves": 4})]:
    qa.set_option("no_hnsw_pq_block", 1)
    for k, v in opts.items(): qa.set_option(k, v)
    try:
        got, scored = graph.search(top, ef, scorer, with_scored=True)
        ms0, l0 = C.c_float(), C.c_uint32(); F.check(lib.qmx_query_timing(scorer._h, C.byref(ms0), C.byref(l0)))
        reps = 3
        for _ in range(reps): got, scored = graph.search(top, ef, scorer, with_scored=True)
        ms, l = C.c_float(), C.c_uint32(); F.check(lib.qmx_query_timing(scorer._h, C.byref(ms), C.byref(l)))
        same = None
        if ref is None: ref = got
        else: same = all(np.array_equal(a, b) for a, b in zip(ref, got))
        print(json.dumps({"variant": name, "kernel_ms": round(ms.value / max(1, l.value), 3), "scored_per_query": round(scored / nq, 1), "kernel": F.last_kernel(scorer._h)[:60],
                          "same_lists_as_uncapped": same}), flush=True)
    finally:
        for k in opts: qa.set_option(k, -1)
        qa.set_option("no_hnsw_pq_block", -1)
PY
timeout 900 python /tmp/pqwalk10m.py 10000000 > gpurun_out/r4d_pqwalk_10m.jsonl 2> gpurun_out/r4d_pqwalk_10m.err
cat gpurun_out/r4d_pqwalk_10m.jsonl; tail -5 gpurun_out/r4d_pqwalk_10m.err
