"""The benchmark line the repository keeps under profiles/ has the fields the bench
contract names, and the derived ones follow from the others (CPU only: it parses the
committed line, it does not run the benchmark)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r3", "bench.json")


@pytest.fixture(scope="module")
def line():
    with open(LINE) as f:
        rows = [r for r in f.read().splitlines() if r.strip()]
    return json.loads(rows[-1])


def test_required_fields(line):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"] == "hamming_compares_per_sec" and line["unit"] == "compares/s"
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["n_gpus"] == 1 and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]


def test_value_follows_from_the_step_time(line):
    cfg = line["config"]
    compares = cfg["nq"] * cfg["nt"] * line["n_gpus"]
    assert line["value"] == pytest.approx(compares / (line["ms_per_step"] * 1e-3), rel=1e-6)


def test_roofline_is_recomputable(line):
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_us", "flop_per_compare"):
        assert k in r, k
    assert r["bound"] in ("mfma", "hbm", "valu-int32")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    cfg = line["config"]
    flops = cfg["nq"] * cfg["nt"] * r["flop_per_compare"]
    assert r["achieved"] == pytest.approx(flops / (r["kernel_us"] * 1e-6) / 1e12, rel=1e-6)      # TFLOP/s
    assert r["traffic"] is None or "traffic_source" in r


def test_cpu_baseline_fields(line):
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    fe = line.get("frontend")
    if fe:
        assert fe["metric"] == "frontend_frames_per_sec" and "roofline" in fe and "cpu_baseline" in fe


def test_headline_duration_is_one_of_the_measured_ones(line):
    """`roofline.kernel_us` is the duration the launch-filled events report (or, where that probe is
    not available, the events recorded around one launch); the other in-run figures stand beside
    it and bracket it: back to back below, recorded events above."""
    r = line["roofline"]
    if "kernel_us_events_around_one_launch" not in r:
        pytest.skip("line written before the dispatch probe existed")
    around = r["kernel_us_events_around_one_launch"]
    by_dispatch = r.get("kernel_us_by_dispatch_events")
    assert r["kernel_us"] in (by_dispatch, around)
    assert r["kernel_us"] <= around
    assert r["kernel_us_back_to_back"] <= around
    assert r["kernel_us"] * 1e-3 <= line["ms_per_step"]          # a launch is shorter than the step it is part of
    assert r["frac_of_sustained_instruction_peak"] == pytest.approx(
        r["achieved"] / r["sustained_instruction_peak"], rel=1e-9)
