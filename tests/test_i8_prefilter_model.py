"""The arithmetic of the int8 copy's prefilter (qdrant_amd/csrc/scan_split.hip, "The INT8 copy"), restated in numpy and checked on the CPU:
the band the device derives from the codes really bounds |exact score - approximate score| - on Gaussian rows and on the data that
stretches it (outlier columns, sparse rows, rows of one repeated value, tiny and huge magnitudes) - and the pass built on it (thresholds from
exact lower bounds, one band) keeps every member of the exact top k.  The device code follows these formulas line by line; the GPU tests
(test_gpu_i8_copy.py) compare its results with the oracle's."""
import zlib

import numpy as np
import pytest


def choose_scales(colmax, colsq, n):
    """split_i8_choose_scales (scan_split.hip), line by line: any s_i >= max |x_i| / 127 keeps the codes in range and |e_i| <= 1/2; raising the scales of
    columns whose typical |q_i s_i| is below 1 / G of the largest moves range from their row codes to their query codes; the G of the smallest
    predicted band (queries distributed like the rows) is taken.  Returns (scales, G) - G = 0: every column at its floor."""
    dim = len(colmax)
    s0 = np.where(colmax > 1e-30, colmax.astype(np.float64) / 127.0, 1.0)
    sig = np.sqrt(colsq.astype(np.float64) / max(n, 1))
    sig = np.where(sig < 1e30, sig, 0.0)
    wmax = float((sig * s0).max()) if dim else 0.0

    def scales_of(G):
        w = sig * s0
        raise_it = (w > 0) & (wmax / G > w) & (colmax > 1e-30)
        return np.where(raise_it, s0 * (wmax / G / np.where(w > 0, w, 1.0)), s0)

    def predicted(s):
        v = sig * s
        t = 3.0 * v.max() / 127.0
        if not t > 0:
            return 0.0
        return 0.4 * v.sum() + t * np.sqrt(((sig / s) ** 2).sum()) * np.sqrt(np.minimum(1.0 / 12.0, (v / t) ** 2).sum())

    best_s, best_b, best_g = None, 0.0, 1e30
    for k, G in enumerate([1e30, 64.0, 45.0, 32.0, 23.0, 16.0, 11.0, 8.0, 5.6, 4.0, 2.8, 2.0]):
        s = scales_of(G)
        b = predicted(s)
        if k == 0 or b < best_b * 0.98:
            best_s, best_b, best_g = s, b, G
    floor_s = np.where(colmax > 1e-30, colmax / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    return np.maximum(best_s.astype(np.float32), floor_s), (0.0 if best_g >= 1e29 else best_g)


def quantize_rows(x, balanced=False):
    colmax = np.abs(x).max(axis=0).astype(np.float32)
    s = np.where(colmax > 1e-30, colmax / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    if balanced:
        s, _ = choose_scales(colmax, (x.astype(np.float32) ** 2).sum(axis=0, dtype=np.float32), x.shape[0])
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
    elif kind == "dominant8":
        x[:, :8] *= 12.0
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    elif kind == "ramp":
        x *= np.linspace(0.05, 8.0, dim, dtype=np.float32)[None, :]
    return x


@pytest.mark.parametrize("kind", ["gauss", "outlier_columns", "sparse", "constant_rows", "tiny", "huge", "zero_column", "dominant8", "ramp"])
@pytest.mark.parametrize("dim", [128, 768])
@pytest.mark.parametrize("balanced", [False, True])
def test_band_bounds_the_error(kind, dim, balanced):
    """... under the floor scales (max |x_i| / 127) and under the balanced ones the segment really takes (choose_scales): the bound holds for ANY scales at
    or above the floor, which is what lets the library pick them for a narrow band."""
    rng = np.random.default_rng(zlib.crc32(kind.encode()) + dim)
    n, nq = 4000, 24
    x = make_rows(kind, rng, n, dim)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    if kind == "sparse":
        q[::2] *= (rng.random((nq // 2, dim)) < 0.1)
    if kind in ("dominant8", "ramp"):
        q[::2] = make_rows(kind, rng, nq // 2, dim)          # half of the queries distributed like the rows
    s, c, c1, c2sq = quantize_rows(x, balanced)
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


def test_balanced_scales_narrow_the_band_where_columns_differ_and_change_nothing_where_they_do_not():
    """unit Gaussian rows with 8 coordinates twelve times the others - with every column at its floor scale the query's one scale is set
    by the 8 large coordinates and the 760 small ones get codes of 0 / +-1: the band is ~1.1 standard deviations of the score and thousands of rows
    fall inside it; balanced, ~0.45 and a few hundred.  Columns of one size (Gaussian rows) keep their floor scales exactly."""
    rng = np.random.default_rng(5)
    n, dim, nq, k = 60_000, 768, 8, 10
    for kind, want_balanced in (("dominant8", True), ("gauss", False)):
        x = make_rows(kind, rng, n, dim)
        q = make_rows(kind, rng, nq, dim)
        exact = (x @ q.T).astype(np.float64)
        sd = exact.std(axis=0)
        kth = np.sort(exact, axis=0)[-k]
        out = {}
        for balanced in (False, True):
            s, c, c1, c2sq = quantize_rows(x, balanced)
            t, d, band = quantize_queries(q, s, c1, c2sq, float(np.linalg.norm(x, axis=1).max()))
            est = (c.astype(np.int64) @ d.astype(np.int64).T).astype(np.float64) * t
            assert (np.abs(est - exact) <= band[None, :]).all()
            out[balanced] = (float(np.mean(band / sd)), float((est >= (kth - band)[None, :]).sum(axis=0).mean()), s)
        colmax = np.abs(x).max(axis=0).astype(np.float32)
        _, G = choose_scales(colmax, (x ** 2).sum(axis=0, dtype=np.float32), n)
        if want_balanced:
            assert G > 0 and out[True][0] < 0.6 * out[False][0] and out[True][1] < 0.25 * out[False][1], (G, out[False][:2], out[True][:2])
        else:
            assert G == 0.0 and np.array_equal(out[True][2], out[False][2])


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
