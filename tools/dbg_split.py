import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
import oracle_ffi as O, qdrant_amd as qa
qa.set_option("debug", 2)
n, dim, nq, top = 300000, 128, 128, 1
rows = O.preprocess(O.COSINE, O.synth(0x5EED0500 + dim, 0, n, dim))
vs = qa.VectorStorage(rows, qa.Distance.Cosine)
st = O.DenseStorage(O.F32, O.COSINE, rows)
for rep in range(12):
    queries = O.synth(0x5EED0501 + nq + rep, 0, nq, dim)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    want = st.peek_top(queries, top, threads=8)
    ok = all(g["idx"].tolist() == w["idx"].tolist() and np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32)) for g, w in zip(got, want))
    print("rep", rep, "ok", ok, flush=True)
