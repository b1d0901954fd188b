"""Mirrors of the reference's PROPERTY tests for the HNSW and PQ halves of the path, run against the oracle (CPU).

The reference holds exactly one literal known-answer test for this half (links_container.rs `test_connect_new_point`, mirrored in
test_oracle_hnsw.py); everything else it tests through properties on data from `StdRng::seed_from_u64(42)` / `SmallRng` (rand 0.10.1,
not reproducible without the crate).  Those properties are restated here with the same shapes and bars on our own seeded data, so the
oracle is held to every bar the reference holds itself to:

  links_container.rs:393-448   test_connect_new_point_with_heuristic   connect_with_heuristic == connect_with_heuristic_simple, 1000 trials
  entry_points.rs:154-177      test_entry_points                       one entry point, `entry_points_num` extra ones
  graph_layers_builder.rs:704-790, 792-905   test_parallel_graph_build / test_add_points   1000 x d=8 cosine, M = 8, ef_construct = 16,
                                             no heuristic: entry level > 0, entry level + 1 == levels, links0 / n > M, search(top 5, ef 16) == exact top 5
  lib/quantization/tests/integration/test_pq.rs:19-320   |PQ score - exact| < 0.05 * dim for dot / l2 / l1, plain and inverted, and score_internal

The GPU counterparts of the builder and PQ properties are in test_gpu_hnsw_build.py / test_gpu_pq.py."""
import numpy as np
import pytest

import oracle_ffi as O


# ---- links_container.rs test_connect_new_point_with_heuristic ----------------------------------------------------------
def _heuristic(cands, m, score):
    """fill_from_sorted_with_heuristic (links_container.rs:47-71), restated independently in Python."""
    links = []
    for idx, sc in cands:
        if any(score(idx, e) > sc for e in links):
            continue
        links.append(idx)
        if len(links) >= m:
            break
    return links


def _connect_simple(links, new, target, m, score):
    """connect_with_heuristic_simple (links_container.rs:107-132): the reference's own reference implementation."""
    if len(links) < m:
        return links + [new]
    cands = [(i, score(target, i)) for i in links] + [(new, score(target, new))]
    # sort_unstable_by(total_cmp) descending: ties are broken arbitrarily in the reference; none occur on continuous data
    cands.sort(key=lambda c: -float(c[1]))
    return _heuristic(cands, m, score)


class _Container:
    """LinksContainer with the order cache (links_container.rs:139-222): the variant the reference ships."""

    def __init__(self):
        self.links, self.processed = [], 0

    def fill(self, cands, m, score):
        self.links = _heuristic(cands, m, score)
        self.processed = len(self.links)

    def connect_with_heuristic(self, new, target, m, score):
        if len(self.links) < m:
            self.links.append(new)
            return
        cache = {}

        def cached(idx):
            if idx not in cache:
                cache[idx] = score(target, idx)
            return cache[idx]
        # NonZeroU32::new(order): order 0 is None, like every unprocessed link
        items = [(l, o if (o < self.processed and o != 0) else None) for o, l in enumerate(self.links)] + [(new, None)]
        import functools

        def cmp(a, b):
            if a[1] is not None and b[1] is not None:
                return -1 if a[1] < b[1] else (1 if a[1] > b[1] else 0)
            sa, sb = float(cached(a[0])), float(cached(b[0]))
            return -1 if sb < sa else (1 if sb > sa else 0)                # b.total_cmp(a): descending
        items.sort(key=functools.cmp_to_key(cmp))
        out = []
        for cand in items:
            skip = False
            for ex in out:
                if cand[1] is not None and ex[1] is not None:
                    continue
                if score(cand[0], ex[0]) > cached(cand[0]):
                    skip = True
                    break
            if skip:
                continue
            out.append(cand)
            if len(out) >= m:
                break
        self.links = [c[0] for c in out]
        self.processed = len(self.links)


def test_connect_with_heuristic_equals_the_simple_variant():
    rng = np.random.default_rng(42)
    NUM, DIM, M = 20, 128, 5
    for trial in range(1000):
        vecs = rng.random((NUM, DIM), dtype=np.float32)
        d = vecs[:, None, :] - vecs[None, :, :]
        t = (-(d * d).sum(-1)).astype(np.float32)                       # Distance::Euclid score_internal = -squared distance
        ids = rng.permutation(NUM).tolist()
        query = ids.pop()
        score = lambda a, b: t[a, b]                                    # noqa: E731
        first = sorted([(i, t[query, i]) for i in ids[:5]], key=lambda c: -float(c[1]))
        ref = _heuristic(first, M, score)
        got = O.links_heuristic(np.array(first, dtype=O.ScoredPointOffset), M, t)
        shipped = _Container()
        shipped.fill(first, M, score)
        assert got == ref == shipped.links
        for c in ids[5:]:
            ref = _connect_simple(ref, c, query, M, score)                # the reference's in-test reference implementation
            got = O.links_connect_heuristic(got, c, query, M, t)          # the oracle (C)
            shipped.connect_with_heuristic(c, query, M, score)            # the variant with the order cache
            assert got == ref == shipped.links, (trial, c)


# ---- graph_layers_builder.rs test_parallel_graph_build / test_add_points, entry_points.rs test_entry_points -------------
@pytest.mark.parametrize("threads", [0, 2])
def test_builder_properties_of_the_reference(threads):
    n, dim, M = 1000, 8, 8
    rows = O.preprocess(O.COSINE, O.synth(0x5EED0700 + threads, 0, n, dim))
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    g = O.Hnsw(st, m=M, ef_construct=16, entry_points_num=10, use_heuristic=False, seed=42, threads=threads)
    ep_ids, ep_lv = g.entry_points()
    xp_ids, _ = g.extra_entry_points()
    assert len(ep_ids) == 1 and len(xp_ids) == 10                        # entry_points.rs:165-166
    assert ep_lv[0] > 0                                                  # main_entry.level > 0
    levels = [g.point_level(i) for i in range(n)]
    assert ep_lv[0] + 1 == max(levels) + 1                               # main_entry.level + 1 == num_levels
    total0 = sum(len(g.links(i, 0)) for i in range(n))
    assert total0 / n > M                                                # total_links_0 / num_vectors > M
    hits = 0
    queries = O.synth(0x5EED0701, 0, 20, dim)
    for q in queries:
        want = st.peek_top(q[None, :], 5)[0]
        got = g.search_dense(st, q[None, :], 5, 16)[0]
        hits += int(got["idx"].tolist() == want["idx"].tolist())
    # the reference asserts equality for its ONE random query; over 20 queries on this small graph every walk is exact too
    assert hits == 20


# ---- lib/quantization/tests/integration/test_pq.rs --------------------------------------------------------------------
VECTORS_COUNT, VECTOR_DIM = 513, 65
ERROR = VECTOR_DIM * 0.05


def _exact(distance, a, b):
    a, b = a.astype(np.float32), b.astype(np.float32)
    if distance == O.DOT:
        return float((a * b).sum(dtype=np.float32))
    if distance == O.EUCLID:
        return float(((a - b) ** 2).sum(dtype=np.float32))                # metrics.rs l2_similarity: sum of squares (no sqrt, no sign)
    return float(np.abs(a - b).sum(dtype=np.float32))


@pytest.mark.parametrize("distance", [O.DOT, O.EUCLID, O.MANHATTAN])
@pytest.mark.parametrize("invert", [False, True])
def test_pq_error_bounds_of_the_reference(distance, invert):
    rng = np.random.default_rng(42 + distance)
    data = rng.random((VECTORS_COUNT, VECTOR_DIM), dtype=np.float32)      # rng.random(): uniform [0, 1)
    query = rng.random(VECTOR_DIM, dtype=np.float32)
    # EncodedVectorsPQ::encode(.., chunk_size = 1, max_kmeans_threads = 1): k-means per 1-d chunk on all 513 vectors
    cen, _ = O.PqOracle.train_ex(data, VECTOR_DIM, 1, 256, max_iters=100, accuracy=1e-5, threads=1)
    pq = O.PqOracle(distance, VECTOR_DIM, 1, cen, invert=invert)
    pq.encode(data)
    scores = pq.score_points(query[None, :], np.arange(VECTORS_COUNT))[0]
    sign = -1.0 if invert else 1.0
    for i in range(VECTORS_COUNT):
        assert abs(scores[i] - sign * _exact(distance, query, data[i])) < ERROR
    if distance == O.DOT:                                                 # test_pq_dot_internal / test_pq_dot_inverted_internal
        internal = pq.score_internal([0] * (VECTORS_COUNT - 1), list(range(1, VECTORS_COUNT)))
        for i in range(1, VECTORS_COUNT):
            assert abs(internal[i - 1] - sign * _exact(distance, data[0], data[i])) < ERROR


def test_byte_storage_ranks_like_the_float_storage():
    """test_byte_storage_hnsw (lib/segment/tests/integration/byte_storage_hnsw_test.rs:39-47, 262-266), its plain-search half for the Uint8 datatype:
    5 000 vectors of 8 byte-valued coordinates (`random_dense_byte_vector`: 0 ..= 255), cosine; the same points in an f32 segment (normalised at insert)
    and in a u8 segment (bytes as they are, the cosine taken per pair); 100 nearest queries of the same kind, top 3: the same ids, scores within 1e-3.
    (Its HNSW half searches under a payload range filter through payload-index sub-graphs: outside the scoring path.)"""
    rng = np.random.default_rng(42)
    n, dim, top = 5000, 8, 3
    raw = np.floor(rng.uniform(0.0, 256.0, (n, dim))).clip(0, 255).astype(np.float32)
    st_f32 = O.DenseStorage(O.F32, O.COSINE, O.preprocess(O.COSINE, raw))
    st_u8 = O.DenseStorage(O.U8, O.COSINE, raw.astype(np.uint8))
    queries = np.floor(rng.uniform(0.0, 256.0, (100, dim))).clip(0, 255).astype(np.float32)
    a = st_f32.peek_top(queries, top)
    b = st_u8.peek_top(queries, top)
    for x, y in zip(a, b):
        assert x["idx"].tolist() == y["idx"].tolist()
        assert np.abs(x["score"] - y["score"]).max() < 1e-3
