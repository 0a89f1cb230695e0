"""The ONE stdout line of bench.py (driver contract): <= 4 KB, parses, carries `roofline` + `cpu_baseline`; everything
else lives in the extras file.  Round 4's line had grown to 20.7 KB and the driver recorded `parsed: null`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _canned():
    """a full result dict of the shape run_rank() builds: round 4's committed 20.7 KB line, with every string padded"""
    res = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    res["config"]["workload"] = res["config"]["workload"] + " x" * 400
    res["dtype"] = res["dtype"] * 3
    res["cpu_baseline"]["sample"] = res["cpu_baseline"]["sample"] * 4
    res["roofline"]["measured"] = "m" * 1000
    res["roofline"]["algorithmic_flop"] = 309237645312
    res["rccl_ranks"] = 8
    res["all_gather"] = {"calls_per_step": 1.0, "bytes_per_step": 17244160, "ms_per_step": 0.2, "gb_per_s": 86.2,
                         "note": "n" * 500}
    res["metric_20k"] = {"distmat_exact_ms": 3.1, "distmat_split3_ms": 1.15, "distmat_f16_ms": 0.54, "distmat_f16_frac": 0.45,
                         "rerank_ms": 4.12, "rerank_vs_cpu_x": 1100.0, "dmap_split": 2.7e-8, "dr1_split": 0.0,
                         "not_a_line_key": "x" * 300}
    return res


def test_line_is_small_parses_and_carries_the_contract(tmp_path):
    res = _canned()
    assert len(json.dumps(res)) > 20000
    text, extras = bench.format_line(res)
    assert "\n" not in text
    assert len(text.encode()) < bench.MAX_LINE_BYTES < 8192
    j = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "extras_file"):
        assert k in j, k
    assert j["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert set(j["config"]) <= set(bench.CONFIG_KEYS) and "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for k in ("bound", "kernel", "shape", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["frac"] <= 1.0
    c = j["cpu_baseline"]
    assert set(c) == {"value", "unit", "cores", "kind", "sample"} and c["kind"] in ("port", "reference")
    assert j["rccl_ranks"] == 8 and set(j["all_gather"]) == set(bench.ALL_GATHER_KEYS)
    # the rest of BASELINE.json's metric (20k x 20k distmat + re-rank ms, parity deltas): bare numbers under one key
    m = j["metric_20k"]
    assert set(m) <= set(bench.METRIC_20K_KEYS) and m["rerank_ms"] == 4.12 and m["distmat_exact_ms"] == 3.1
    assert all(isinstance(v, (int, float)) or v is None for v in m.values())
    assert bench.MAX_STR <= 110   # the driver's record clipped strings near 120 characters in round 5

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(s_) for s_ in strings(j)) <= bench.MAX_STR
    # nothing is lost: the long material is in the extras
    for k in ("roofline_all", "extras", "gemm_classes", "reference_cpu", "drop_in"):
        assert k in extras, k
    assert extras["roofline"]["measured"] == "m" * 1000


def test_emit_writes_extras_then_prints_one_line(tmp_path, capsys):
    res = _canned()
    path = str(tmp_path / "bench_extras.json")
    bench.emit(res, path)
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["extras_file"] == path
    assert "roofline_all" in json.load(open(path))


def test_gpus_beyond_visible_devices_exit_2_with_a_message():
    """`--gpus 8` where fewer devices are visible: a clear message and exit code 2 before anything touches HIP or joins
    a rendezvous -- in the self-launching parent and in a rank started by torch.distributed.run alike"""
    import torch
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MPREID_DIST_BACKEND",
                                                             "MPREID_ALLOW_SHARED_GPU")}
    n = have + 1 if have else 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--small", "--steps", "1",
                        "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert f"--gpus {n} but only {have} GPU(s) are visible" in r.stderr
    env.update(WORLD_SIZE=str(n), RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--small"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and r.stdout.strip() == "" and r.stderr.strip() == ""   # only local rank 0 speaks


@pytest.mark.gpu
def test_bench_on_the_gpu_prints_one_small_line_and_a_quiet_stderr(tmp_path):
    """the driver's view of `python bench.py --gpus 1 ...` (a 1/16 workload): stdout is ONE line that parses and obeys the size
    contract, stderr is a handful of lines (round 4: ~90 evaluator lines), the extras file lands in the current directory"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--small", "--steps", "2", "--warmup", "1",
                        "--cpu-images", "8"], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) <= bench.MAX_LINE_BYTES
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["value"] > 0 and j["unit"] == "images/s"
    assert j["roofline"]["bound"] == "mfma" and 0 < j["roofline"]["frac"] <= 1 and j["roofline"]["avg_launch_ms"] > 0
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] >= 1
    assert j["config"]["encoder_precision"] == "split" and "Market-1501" in j["config"]["workload"]
    assert "HBM-resident loader" in j["config"]["workload"]

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert not [s_ for s_ in strings(j) if s_.endswith("...") or len(s_) > bench.MAX_STR], "a string of the line was clipped"
    # the parity clause of the metric, measured live against the CPU checker on the cpu_baseline leg's sample
    m = j["metric_20k"]
    assert m["parity_images"] >= 8 and m["feat_rel_l2_split"] <= 2e-5 and m["dist_err_exact"] == 0.0 and m["dist_err_split3"] <= 1e-5
    assert j["roofline"]["traffic_replayed"] in (True, False, None)
    assert len([ln for ln in r.stderr.splitlines() if ln.strip()]) <= 12, r.stderr[-1500:]
    extras = json.load(open(tmp_path / "bench_extras.json"))
    assert "gemm_classes" in extras and "drop_in" in extras
