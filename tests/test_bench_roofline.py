"""bench.py's roofline arithmetic for the kernels it may name as dominant (host logic, no GPU): the bytes it prices a launch at, the roof it picks,
the PMC traffic it attaches from profiles/pmc_traffic.json - and that the bench lines committed under profiles/ are consistent with it."""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, DIM, Q = 10_000_000, 768, 128
I8 = "qmx::scan_i8copy_kernel(qmx::ScanArgs, qmx::SplitArgs)"
HALF = "void qmx::scan_f16pair_kernel<true>(qmx::ScanArgs, qmx::SplitArgs)"
EXACT16 = "void qmx::scan_f32_mfma16_kernel<6, 4, 1, 0, false, false>(qmx::ScanArgs)"


@pytest.mark.parametrize("symbol,elem_bytes,peak,dtype", [(I8, 1, bench.MFMA_I8_PEAK_TOPS, "int8"), (HALF, 2, bench.MFMA_F16_PEAK_TFLOPS, "f16")])
def test_prefilter_kernels_are_priced_on_the_bytes_of_their_copy(symbol, elem_bytes, peak, dtype):
    kernel_ms, launches_per_pass = 0.644 * elem_bytes, 2.0
    alg = int(N * DIM * elem_bytes / launches_per_pass)                     # the pass's two launches cover the copy once: the mean launch
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    r = bench._roofline(N, DIM, kernel_ms, alg, achieved, 200, 2.0, Q, symbol, launches_per_pass)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == bench.HBM_PEAK_GBPS
    assert r["frac"] == pytest.approx(achieved / 8000.0, abs=1e-4) and r["achieved"] == pytest.approx(achieved, abs=0.1)
    assert r["queries_per_pass"] == 128.0 and r["launches_per_pass"] == 2.0
    assert r["algorithmic_bytes_per_launch"] == alg
    assert r["mfma"]["peak_TFLOPs"] == peak and r["mfma"]["dtype"].startswith(dtype)
    assert r["algorithmic_flops_per_launch"] == 2.0 * N * DIM * 128 / 2.0   # one product per element of a padded 128-query tile
    # PMC traffic of THIS symbol at THIS row count, per launch like `achieved`
    assert r["traffic"] is not None and r["traffic_over_algorithmic"] == pytest.approx(1.002, abs=0.002)
    assert (dtype + " copy") in r["f32_block_equivalent"]["note"]
    # a stale number is worse than none: another row count carries no traffic
    assert bench._roofline(N // 2, DIM, kernel_ms, alg // 2, achieved, 200, 2.0, Q, symbol, launches_per_pass)["traffic"] is None


def test_exact_scan_is_priced_on_the_f32_block():
    kernel_ms = 4.46
    alg = N * DIM * 4
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    r = bench._roofline(N, DIM, kernel_ms, alg, achieved, 100, 1.0, 16, EXACT16)
    assert r["bound"] == "hbm" and r["frac"] == pytest.approx(0.861, abs=2e-3)
    assert r["mfma"]["dtype"] == "f32" and r["mfma"]["peak_TFLOPs"] == bench.MFMA_F32_PEAK_TFLOPS
    assert "f32_block_equivalent" not in r


@pytest.mark.parametrize("name", ["r3_bench_c2_i8_default.json", "r3_bench_full_default.json"])
def test_committed_bench_lines_are_self_consistent(name):
    d = json.loads(open(os.path.join(ROOT, "profiles", name)).read())
    assert d["unit"] == "queries/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["batch"] == 128 and d["config"]["rows_per_gpu"] == N and d["config"]["dim"] == DIM
    assert d["value"] == pytest.approx(d["config"]["batch"] / (d["ms_per_step"] * 1e-3), rel=1e-3)
    r = d["roofline"]
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=1e-4)
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9, rel=2e-3)
    assert r["traffic"] == pytest.approx(r["algorithmic_bytes_per_launch"], rel=5e-3)
    # two scan launches per 128-query step; the rest of the step is what the timeline files itemise
    assert 0.0 < d["ms_per_step"] - 2 * r["kernel_ms"] < 0.5
    assert d["prefilter_equals_exact_scan_whole_block"] is True and d["recall_at_10"] == 1.0
    for leg in d["robustness"].values():
        assert leg["equals_exact_scan_whole_block"] is True and leg["fallback_rate"] == 0.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    if "half_copy_point" in d:
        assert d["half_copy_point"]["equals_exact_scan_whole_block"] is True and "scan_f16pair_kernel<true>" in d["half_copy_point"]["kernel"]
        assert "scan_i8copy_kernel" in r["kernel"]


def test_timed_path_names_what_ran():
    assert bench._timed_path(I8).startswith("prefilter over an int8 copy") and "bit for bit" in bench._timed_path(I8)
    assert bench._timed_path(HALF).startswith("prefilter over an f16 copy")
    assert bench._timed_path("void qmx::scan_f16pair_kernel<false>(qmx::ScanArgs, qmx::SplitArgs)").startswith("prefilter over an f16-pair copy")
    assert bench._timed_path("qmx::scan_f32_split_kernel(qmx::ScanArgs, qmx::SplitArgs)").startswith("prefilter converting")
    assert bench._timed_path(EXACT16) == "exact f32 scan"
