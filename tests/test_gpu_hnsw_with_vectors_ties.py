"""`search_on_level_with_vectors` among EQUAL link scores (graph_layers.rs:336-389, search_context.rs:23-40): the reference keeps every evicted candidate in
`candidates` and expands each one whose score is not below the bound (:354-365: the loop breaks on strict `<` only).  Integer link scores (binary
quantization, 1-bit TurboQuant) evict dozens of equal scores at once; the device keeps the latest-score group of evicted candidates in four registers + a
per-slot stack (hnsw.hpp, round 6; rounds 4 - 5 kept four and dropped the fifth silently).

What is pinned here: the device walk == the reference's ALGORITHM run with the device's documented order among equal scores (key = score, then the lower id
first, for `nearest` and for `candidates` - where the reference's order is that of its two binary heaps: DESIGN 4).  The model below is that algorithm in
plain Python over the oracle's link scores; the test asserts that the searches really evict eight and more equal scores at a time."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


def _key(score, idx):
    return (float(score), -int(idx))           # larger = better: the higher score, then the lower id


def model_walk(links_of, entry, ef, limit, link_scores):
    """-> (popped ids in order, the break candidate included; scored link count; the largest number of equal-score evicted candidates waiting at once)"""
    visited = {entry}
    nearest = [(_key(link_scores[entry], entry), entry)]        # SearchContext::nearest as a sorted list (best first), at most ef
    cands = {entry: _key(link_scores[entry], entry)}            # SearchContext::candidates
    in_nearest_unexpanded = {entry}
    evicted = {}                                                # evicted and not yet popped: id -> key
    popped, n_scored, worst_tie = [], 1, 0
    while cands:
        cid = max(cands, key=cands.get)
        ck = cands.pop(cid)
        lower = nearest[-1][0][0]                               # lower_bound(): the worst of nearest, full or not (search_context.rs:23-28)
        popped.append(cid)
        in_nearest_unexpanded.discard(cid)
        evicted.pop(cid, None)
        if ck[0] < lower:
            break
        fresh = [l for l in links_of(cid) if l not in visited][:limit]
        for l in fresh:
            n_scored += 1
            visited.add(l)
            k = _key(link_scores[l], l)
            if len(nearest) < ef:
                nearest.append((k, l))
            elif k > nearest[-1][0]:
                out = nearest.pop()[1]
                if out in in_nearest_unexpanded:                # it stays in `candidates`: the reference never removes it from the heap
                    in_nearest_unexpanded.discard(out)
                    evicted[out] = cands[out]
                nearest.append((k, l))
            else:
                continue                                        # rejected by nearest.push: not a candidate (search_context.rs:33-39)
            nearest.sort(key=lambda t: t[0], reverse=True)
            cands[l] = k
            in_nearest_unexpanded.add(l)
        bound = nearest[-1][0][0]
        worst_tie = max(worst_tie, sum(1 for k in evicted.values() if k[0] == bound))
    return popped, n_scored, worst_tie


def _level0_only(p, n):
    end = int(p.offsets[n])
    return O.PlainLinks(p.m, p.m0, p.reindex, np.array([0, n], dtype=np.uint64), p.offsets[:n + 1].copy(), p.neighbors[:end].copy(),
                        np.array([p.ep_ids[0]], dtype=np.uint32), np.array([0], dtype=np.uint32))


@pytest.mark.parametrize("kind,dim,ef", [("bq", 64, 16), ("bq", 128, 48), ("tq1", 64, 24)])
def test_search_with_vectors_expands_every_evicted_candidate_that_ties_with_the_bound(kind, dim, ef):
    import qdrant_amd as qa
    n, m, nq, top = 3000, 16, 12, 10
    distance = O.DOT
    rng = np.random.default_rng(dim * 7 + ef)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    st = O.DenseStorage(O.F32, distance, rows)
    p = _level0_only(O.Hnsw(st, m=m, ef_construct=40, seed=5).export_plain(), n)
    graph = qa.GraphLayers.from_plain(p)
    vs = qa.VectorStorage(rows, qa.Distance.Dot)
    if kind == "bq":
        oq = O.BqOracle(distance, dim)
        oq.encode_rows(rows)
        quant = qa.BinaryQuantizer(dim, qa.Distance.Dot)
        qs = qa.EncodedVectorsBin(quant.encode(rows), quant)
        link = oq.score_points(queries, np.arange(n))
    else:
        oq = O.TqOracle(distance, dim, 3)                       # TQBits::Bits1
        quant = qa.TurboQuantizer(dim, qa.Distance.Dot, 3)
        qs = qa.EncodedVectorsTQ(oq.encode_rows(rows), quant)
        link = oq.score_points(queries, np.arange(n))
    links_scorer, base_scorer = qa.new_raw_scorer(queries, qs), qa.new_raw_scorer(queries, vs)
    # the model's link scores are the device's, bit for bit (the scorers' own parity tests hold this; asserted so that a difference below is the walk's)
    dev_link = links_scorer.score_points(np.arange(n, dtype=np.uint32))
    assert np.array_equal(np.asarray(dev_link).view(np.uint32), link.view(np.uint32))
    got, scored = graph.search_with_vectors(top, ef, links_scorer, base_scorer, with_scored=True)
    base = st.score_points(queries, np.arange(n))
    entry, ties, total = int(p.ep_ids[0]), [], 0
    for qi in range(nq):
        popped, n_links, worst_tie = model_walk(lambda c: p.links(c, 0).tolist(), entry, max(ef, top), p.m0, link[qi])
        ties.append(worst_tie)
        total += n_links + len(popped)
        # base_search_context: FixedLengthPriorityQueue(ef) over the base scores of everything popped, into_iter_sorted().take(top)
        order = sorted(popped, key=lambda i: (-float(base[qi, i]), i))[:top]
        assert got[qi]["idx"].tolist() == order, (qi, worst_tie)
        assert np.array_equal(got[qi]["score"].view(np.uint32), base[qi, order].view(np.uint32))
    assert scored == total
    if kind == "bq":                     # (1-bit TurboQuant scores carry a per-row scale: they tie rarely - the case pins the walk itself)
        assert max(ties) >= 8, ties      # eight and more equal scores waited outside `nearest` at once: beyond what four registers hold
