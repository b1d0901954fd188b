"""The arithmetic of the int8 copy's prefilter (qdrant_amd/csrc/scan_split.hip, "The INT8 copy"), restated in numpy and checked on the CPU:
the band the device derives from the codes really bounds |exact score - approximate score| - on Gaussian rows and on the data that
stretches it (outlier columns, sparse rows, rows of one repeated value, tiny and huge magnitudes) - and the pass built on it (thresholds from
exact lower bounds, one band) keeps every member of the exact top k.  The device code follows these formulas line by line; the GPU tests
(test_gpu_i8_copy.py) compare its results with the oracle's."""
import zlib

import numpy as np
import pytest


def quantize_rows(x):
    colmax = np.abs(x).max(axis=0).astype(np.float32)
    s = np.where(colmax > 1e-30, colmax / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    c = np.clip(np.rint(x / s), -127, 127).astype(np.int32)
    c1 = int(np.abs(c).sum(axis=1).max())
    c2sq = int((c.astype(np.int64) ** 2).sum(axis=1).max())
    return s, c, c1, c2sq


def quantize_queries(q, s, c1, c2sq, row_norm_max):
    dim = q.shape[1]
    v = (q * s).astype(np.float32)
    mx = np.abs(v).max(axis=1)
    t = np.where(mx > 1e-30, mx / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    xs = (v / t[:, None]).astype(np.float32)
    d = np.clip(np.rint(xs), -127, 127).astype(np.float32)
    f = xs - d
    sabs = np.abs(d).sum(axis=1)
    sf2 = (f.astype(np.float64) ** 2).sum(axis=1)
    cf = np.minimum(0.5 * c1, np.sqrt(c2sq * sf2))
    qn = np.sqrt((q.astype(np.float64) ** 2).sum(axis=1))
    band = t * (cf + 0.5 * sabs + 0.25 * dim) * 1.001 + 2.0 * dim * 1.1920929e-7 * row_norm_max * qn
    return t, d.astype(np.int32), band


def make_rows(kind, rng, n, dim):
    x = rng.standard_normal((n, dim)).astype(np.float32)
    if kind == "gauss":
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    elif kind == "outlier_columns":
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        x[:, :3] *= 40.0
    elif kind == "sparse":
        x *= (rng.random((n, dim)) < 0.05)
    elif kind == "constant_rows":
        x = np.repeat(rng.standard_normal((n, 1)).astype(np.float32), dim, axis=1)
    elif kind == "tiny":
        x *= np.float32(1e-18)
    elif kind == "huge":
        x *= np.float32(1e15)
    elif kind == "zero_column":
        x[:, 5] = 0.0
    return x


@pytest.mark.parametrize("kind", ["gauss", "outlier_columns", "sparse", "constant_rows", "tiny", "huge", "zero_column"])
@pytest.mark.parametrize("dim", [128, 768])
def test_band_bounds_the_error(kind, dim):
    rng = np.random.default_rng(zlib.crc32(kind.encode()) + dim)
    n, nq = 4000, 24
    x = make_rows(kind, rng, n, dim)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    if kind == "sparse":
        q[::2] *= (rng.random((nq // 2, dim)) < 0.1)
    s, c, c1, c2sq = quantize_rows(x)
    row_norm_max = float(np.sqrt((x.astype(np.float64) ** 2).sum(axis=1).max()))
    t, d, band = quantize_queries(q, s, c1, c2sq, row_norm_max)
    acc = c.astype(np.int64) @ d.astype(np.int64).T                     # what the matrix cores deliver, exactly
    est = acc.astype(np.float64) * t.astype(np.float64)
    exact = x.astype(np.float64) @ q.astype(np.float64).T
    f32 = (x @ q.T).astype(np.float64)                                  # an f32 evaluation (the exact scores the device compares with carry this round-off)
    err = np.maximum(np.abs(exact - est), np.abs(f32 - est))
    assert (err <= band[None, :]).all(), float((err / band[None, :]).max())
    if kind == "gauss" and dim == 768:                                  # ... and is the worst case, not the typical one: two orders of room
        assert (err / band[None, :]).max() < 0.2


@pytest.mark.parametrize("top", [1, 10, 64])
def test_the_pass_keeps_the_exact_top_k(top):
    """sample -> T0; first sixteenth with thr = T0 - band; the 64 best candidates' exact scores -> T1; the rest with thr = T1 - band; again -> T2;
    keep approximate >= T2 - band; exact scores of those, top k: the exact scan's list."""
    rng = np.random.default_rng(7 + top)
    n, dim, nq = 60_000, 128, 8
    x = make_rows("gauss", rng, n, dim)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    s, c, c1, c2sq = quantize_rows(x)
    t, d, band = quantize_queries(q, s, c1, c2sq, float(np.linalg.norm(x, axis=1).max()))
    exact = (x @ q.T).astype(np.float32)
    est = (c.astype(np.int64) @ d.astype(np.int64).T).astype(np.float32) * t
    first = (np.arange(n) // 256) % 16 == 0                            # the strided sixteenth of the 256-row tiles
    verified = []
    for j in range(nq):
        sample = np.arange(0, n, n // 1024)
        T = np.sort(exact[sample, j])[-top]
        cand = np.zeros(n, dtype=bool)
        for part in (first, ~first):
            thr = (T - band[j]) / t[j]
            cand |= part & ~((est[:, j] / t[j]) < thr)
            ids = np.flatnonzero(cand)
            probe = ids[np.argsort(-est[ids, j], kind="stable")[:64]]
            if len(probe) >= top:
                T = max(T, np.sort(exact[probe, j])[-top])
        keep = np.flatnonzero(cand & ~(est[:, j] < T - band[j]))
        verified.append(len(keep))
        got = keep[np.lexsort((keep, -exact[keep, j]))[:top]]
        want = np.lexsort((np.arange(n), -exact[:, j]))[:top]
        assert got.tolist() == want.tolist()
    assert max(verified) < 2048
