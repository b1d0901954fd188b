"""bench.py's output contract and launcher (host logic, CPU): the LAST stdout line is a compact headline (<= 4 KB) that carries every key the driver
parses + `roofline` (top level = the SURVEY 8(d) block stream, with `block_stream` and `timed_kernel` under explicit names) + `cpu_baseline`; `--gpus N`
starts its own N ranks when no launcher did and refuses to run on fewer devices than asked."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import bench

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline"]


def _canned():
    """a full result of a real run (the round-4 line, 23.8 KB: the one the driver could not parse) + the keys added since"""
    d = json.loads(open(os.path.join(ROOT, "profiles", "r4_bench_full_default.json")).read())
    d["roofline_hbm_point_q1"] = dict(d["roofline_hbm_point_q16"], batch=1, kernel="void qmx::scan_kernel<qmx::RowF32<0>, 1, 4, 4, false, 0>(qmx::ScanArgs)",
                                      kernel_ms=4.63, achieved=6635.0, frac=0.8294)
    d["collective"] = "none (one GPU)"
    return d


def test_headline_of_a_full_run_fits_4k_and_carries_the_contract():
    full = _canned()
    assert len(json.dumps(full)) > 20000
    h = bench.headline(full, os.path.join(ROOT, "bench_details.json"))
    line = json.dumps(h)
    assert len(line) <= bench.HEADLINE_MAX_BYTES == 4096, len(line)
    for k in CONTRACT:
        assert k in h, k
    assert h["value"] == full["value"] and h["ms_per_step"] == full["ms_per_step"] and h["steps"] == full["steps"] and h["n_gpus"] == 1
    assert h["config"]["workload"].startswith("C2: 1 segment 10M x d=768 f32 cosine") and h["config"]["batch"] == 128
    assert h["details"] == "bench_details.json"
    # nothing was dropped to make it fit
    for k in ("batch_sweep", "robustness", "one_process_fanout", "configs", "checks"):
        assert isinstance(h[k], dict), k
    assert set(h["configs"]) == {"C3", "TQ4", "C4", bench.LEG_COLUMNS}
    legs = h["configs"][bench.LEG_COLUMNS]
    assert {"C3.scan_Q32", "C3.walk", "TQ4.scan_Q32", "C4.walk", "C4.scan_Q32"} <= set(legs) and all(len(v) == 6 for v in legs.values())
    assert legs["C3.walk"][0].startswith("walk<HopRow<RowSQ") and 0 < legs["C3.walk"][3] < 1
    assert h["configs"]["C3"]["oracle_walk"]["default_walk"].startswith("ids 254/256")
    assert h["cpu_baseline"]["kind"] == "port" and h["cpu_baseline"]["cores"] >= 1 and h["cpu_baseline"]["value"] > 0


def test_roofline_top_level_is_the_block_stream_and_both_fractions_are_named():
    full = _canned()
    r = bench.headline(full)["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "block_stream", "block_stream_q1", "timed_kernel"):
        assert k in r, k
    bs, tk = r["block_stream"], r["timed_kernel"]
    # 3072 B per row of the stored block, 10 M rows, streamed once for 16 queries
    assert bs["batch"] == 16 and bs["algorithmic_bytes"] == 30_720_000_000 and "scan_f32_mfma16_kernel" in bs["kernel"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["frac"] == bs["frac"] == pytest.approx(bs["algorithmic_bytes"] / (bs["kernel_ms"] * 1e-3) / 1e9 / 8000.0, abs=2e-3)
    assert r["achieved"] == pytest.approx(r["frac"] * r["peak"], rel=1e-3) and r["achieved"] == pytest.approx(bs["algorithmic_bytes"] / (bs["kernel_ms"] * 1e-3) / 1e9, rel=1e-3)
    assert r["traffic"] == pytest.approx(bs["algorithmic_bytes"], rel=5e-3) and bs["traffic_over_algorithmic"] == pytest.approx(1.0, abs=5e-3)
    assert r["block_stream_q1"]["batch"] == 1 and "scan_kernel<RowF32" in r["block_stream_q1"]["kernel"]
    # the timed kernel on the bytes IT streams (the int8 copy: 7.68 GB per pass, two launches)
    assert "scan_i8copy_kernel" in tk["kernel"] and tk["mfma_frac"] < 1 and tk["bytes_streamed"] == 3_840_000_000 and tk["launches_per_pass"] == 2.0
    assert tk["frac"] == pytest.approx(tk["bytes_streamed"] / (tk["kernel_ms"] * 1e-3) / 1e9 / 8000.0, abs=2e-3)
    assert tk["frac"] < 1.0 and r["frac"] < 1.0


def test_without_a_block_stream_point_the_top_level_is_the_timed_kernel():
    full = _canned()
    del full["roofline_hbm_point_q16"]
    r = bench.headline(full)["roofline"]
    assert "block_stream" not in r and r["frac"] == full["roofline"]["frac"] and r["of"].startswith("timed_kernel")


def test_an_oversized_result_drops_optional_groups_not_contract_keys():
    full = _canned()
    full["batch_sweep"] = {("Q%d_exact" % i): dict(full["batch_sweep"]["Q1_exact"]) for i in range(200)}
    h = bench.headline(full)
    assert len(json.dumps(h)) <= 4096 and h["batch_sweep"] == "see details" and isinstance(h["configs"], dict)
    for k in CONTRACT:
        assert k in h


def test_short_kernel_names():
    assert bench._short("void qmx::scan_f32_mfma16_kernel<6, 4, 1, 0, false, false>(qmx::ScanArgs)") == "scan_f32_mfma16_kernel<6,4,1,0,false,false>"
    assert bench._short("qmx::scan_i8copy_kernel(qmx::ScanArgs, qmx::SplitArgs)") == "scan_i8copy_kernel"
    assert bench._short("void qmx::hnsw_search_kernel<qmx::HopRow<qmx::RowSQ<false, false> >, 2, true>(qmx::ScanArgs, qmx::HnswArgs)").startswith("hnsw_search_kernel<HopRow<RowSQ")


# ---- launcher --------------------------------------------------------------------------------------------------------
def test_resolve_world_never_runs_on_fewer_devices_than_asked():
    a = bench.parse(["--gpus", "8"])
    kind, cmd = bench.resolve_world(a, ["--gpus", "8"], environ={}, device_count=8)
    assert kind == "exec" and cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd
    assert cmd[-2:] == ["--gpus", "8"] and os.path.basename(cmd[-3]) == "bench.py"
    assert bench.resolve_world(a, [], environ={}, device_count=1)[0] == "fail"                                   # one visible GPU, eight asked: refuse
    assert bench.resolve_world(a, [], environ={"WORLD_SIZE": "1", "RANK": "0"}, device_count=8)[0] == "fail"    # launcher disagrees with --gpus
    assert bench.resolve_world(a, [], environ={"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3", "LOCAL_WORLD_SIZE": "8"}, device_count=8) == ("run", 3, 3, 8)
    assert bench.resolve_world(a, [], environ={"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3", "LOCAL_WORLD_SIZE": "8"}, device_count=4)[0] == "fail"
    one = bench.parse([])
    assert bench.resolve_world(one, [], environ={}, device_count=1) == ("run", 0, 0, 1)
    assert bench.resolve_world(one, [], environ={"WORLD_SIZE": "2", "RANK": "0"}, device_count=2)[0] == "fail"
    assert bench.resolve_world(bench.parse(["--gpus", "2", "--backend", "gloo"]), [], environ={}, device_count=0)[0] == "fail"   # gloo is for the test backend only


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="would start a real 2-GPU run")
def test_gpus_2_fails_loudly_where_two_devices_are_not_visible():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert p.returncode == 2, (p.returncode, p.stderr[-400:])
    assert "refusing to run on fewer GPUs than asked" in p.stderr and p.stdout.strip() == ""


def test_gpus_2_launches_its_own_ranks_and_reports_two(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it: bench.py re-executes itself under torch.distributed.run, two ranks meet over gloo, the
    injected oracle backend scores, product code (qdrant_amd.sharded) gathers and merges; the last stdout line says n_gpus 2 / rccl_ranks 2 and the merged
    lists are the oracle's over the union of the two segments."""
    import oracle_ffi as O
    n, dim, Q, nq, steps, top = 900, 40, 4, 8, 3, 10
    details = str(tmp_path / "details.json")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = HERE + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--test-backend", "bench_test_backend:make",
                        "--rows", str(n), "--dim", str(dim), "--batch", str(Q), "--nqueries", str(nq), "--steps", str(steps), "--warmup", "1", "--top", str(top),
                        "--details", details], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "starting 2 ranks" in p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    h = json.loads(lines[-1])
    assert len(lines[-1]) <= 4096
    assert h["n_gpus"] == 2 and h["rccl_ranks"] == 2 and h["collective"] == "gloo" and h["steps"] == steps and h["scaling"] == "weak"
    assert h["config"]["workload"].startswith("C5: 2 segments") and "not a measurement" in h["data"]
    assert "2 x %d vecs" % n in h["metric"]                   # the collection the queries are answered over
    # N > 1 is the experiment of N = 1: the same four batches in flight, + ONE collective per step (lists and counts travel in one packed record)
    assert h["config"]["batches_in_flight"] == 4 and h["config"]["collectives_per_step"] == 1.0
    # (a fixed count on every rank, derived from --prewarm-ms: a step holds a collective, a timed loop would desynchronise the ranks)
    assert h["config"]["prewarm_steps"] == 100
    # value = the COLLECTION's queries per second (not x world): ideal weak scaling reads value(N) == value(1)
    assert h["unit"] == "queries/s" and h["value"] == h["config"]["collection_qps"]
    assert h["value"] == pytest.approx(Q * steps / (h["ms_per_step"] * 1e-3 * steps), rel=1e-2)
    assert h["config"]["segment_searches_per_s"] == pytest.approx(2 * h["value"], abs=0.05)
    full = json.load(open(details))
    assert full["lanes_used"] == [0, 1, 2]          # (3 steps: the first three of the four lanes)
    # the merged lists of the last step = the oracle's exact search over the union of the two segments (ids globalised by the segment bases)
    seed = 0x5EED0002
    rows = np.concatenate([O.preprocess(O.COSINE, O.synth(seed + 16 * r, 0, n, dim)) for r in range(2)])
    b = (steps - 1) % (nq // Q)
    want = O.DenseStorage(O.F32, O.COSINE, rows).peek_top(O.synth(seed + 1, 0, nq, dim)[b * Q:(b + 1) * Q], top)
    checksum = sum(int(w["idx"].view(np.int32).astype(np.int64).sum()) + int(w["score"].view(np.int32).astype(np.int64).sum()) for w in want)
    assert full["merged_checksum"] == checksum


@pytest.mark.parametrize("name", ["r6_bench", "r6_bench_driver_form"])
def test_the_committed_round6_headline_is_what_headline_makes_of_the_committed_details(name):
    """profiles/r6_bench_headline.json is the last stdout line of the round's final `python bench.py`, profiles/r6_bench_details.json the full result of the same
    run: the line must be reproducible from the details, fit 4 KB with nothing dropped, and carry the SURVEY 8(d) block stream as its top-level roofline."""
    # (r6_bench: `python bench.py`, 100 steps; r6_bench_driver_form: the command as the driver runs it, `python3 bench.py --gpus 1 --steps 20 --warmup 5`, final bench.py)
    line = open(os.path.join(ROOT, "profiles", name + "_headline.json")).read().strip().splitlines()[-1]
    h = json.loads(line)
    full = json.load(open(os.path.join(ROOT, "profiles", name + "_details.json")))
    assert len(line) <= 4096
    assert bench.headline(full, os.path.join(ROOT, "bench_details.json")) == h
    for k in CONTRACT:
        assert k in h, k
    for k in ("batch_sweep", "robustness", "one_process_fanout", "configs", "checks"):
        assert isinstance(h[k], dict), k
    assert h["n_gpus"] == 1 and h["rccl_ranks"] == 1 and h["vs_baseline"] is None and h["dtype"].startswith("f32 (int8 prefilter")
    assert h["value"] == pytest.approx(h["config"]["batch"] / (h["ms_per_step"] * 1e-3), rel=1e-3)
    r = h["roofline"]
    assert r["bound"] == "hbm" and r["frac"] == r["block_stream"]["frac"] >= 0.70          # north_star: >= 70 % of the HBM roofline on C2
    assert r["block_stream"]["algorithmic_bytes"] == 30_720_000_000 and r["block_stream"]["traffic_over_algorithmic"] == pytest.approx(1.0, abs=5e-3)
    assert r["timed_kernel"]["kernel"] == "scan_i8copy_kernel" and 0.5 < r["timed_kernel"]["frac"] < 1.0
    assert h["checks"] == {"prefilter_equals_exact_scan_whole_block": True, "recall_at_10": 1.0}
    assert h["cpu_baseline"]["gpu_matches_oracle_on_sample_bit_exact"] is True and h["robustness"]["every_list_equals_exact_scan"] is True
    for c in ("C3", "C4"):
        ow = h["configs"][c]["oracle_walk"]
        assert ow["reference_heap_order"] == "ids 256/256, bits 256/256, pops 256/256", (c, ow)
        n, d = ow["tie_explained"].split("/")
        assert n == d and "unexplained" not in ow
