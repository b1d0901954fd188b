"""BASELINE.json configs[0] ("C1") at its stated size on the device: 100 000 x 128 f32 cosine, brute-force exact top-10, 1 024 queries in batches of
Q in {1, 8, 32} through the C-ABI == the oracle's `peek_top_iter`: ids and score BITS (lib/segment/benches/vector_search.rs:21,34-104,
tests/integration/exact_search_test.rs:165-236)."""
import numpy as np
import pytest

import oracle_ffi as O
from test_oracle_c1 import N, DIM, NQ, TOP, c1_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c1():
    import qdrant_amd as qa
    rows, queries = c1_inputs()
    want = O.DenseStorage(O.F32, O.COSINE, rows).peek_top(queries, TOP, threads=8)
    return qa, qa.VectorStorage(rows, qa.Distance.Cosine), queries, want


@pytest.mark.parametrize("Q,count", [(32, NQ), (8, 256), (1, 64)])
def test_c1_device_equals_oracle_bits(c1, Q, count):
    qa, st, queries, want = c1
    assert st.total_vector_count() == N and st.dim == DIM
    for q0 in range(0, count, Q):
        got = qa.BatchFilteredSearcher(queries[q0:q0 + Q], st, TOP).peek_top_all()
        for j, g in enumerate(got):
            w = want[q0 + j]
            assert g["idx"].tolist() == w["idx"].tolist(), (Q, q0 + j)
            assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32)), (Q, q0 + j)


def test_c1_with_the_int8_copy_returns_the_same_lists(c1):
    """A C1 segment created with QMX_SEG_I8_COPY (the headline's flag), 128 queries per pass: blocks below 2^18 rows are served by the exact kernels (the
    prefilter's fixed costs are not worth 51 MB), and the lists are the oracle's bits either way"""
    qa, _, queries, want = c1
    from qdrant_amd import _ffi as F
    rows, _ = c1_inputs()
    st8 = qa.VectorStorage(rows, qa.Distance.Cosine, flags=F.SEG_I8_COPY)
    for q0 in range(0, 512, 128):
        s = qa.BatchFilteredSearcher(queries[q0:q0 + 128], st8, TOP)
        got = s.peek_top_all()
        assert "scan_f32_mfma16_kernel" in F.last_kernel(s.scorer._h) and s.counters.prefilter_queries == 0
        for j, g in enumerate(got):
            w = want[q0 + j]
            assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32)) and g["idx"].tolist() == w["idx"].tolist()
