"""The bench line the driver parses: built by bench.compact_line from the full record, always below 4 KiB, always carrying the
contract's keys, the dominant kernel's roofline and the CPU baseline (a 25.7 KB line was not recovered by the driver)."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU = ("value", "unit", "cores", "kind", "sample")


def _records():
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_ntt_2_24.json")) + glob.glob(os.path.join(ROOT, "profiles", "r*_bench_detail.json")))
    assert hits
    return hits


@pytest.mark.parametrize("path", _records(), ids=os.path.basename)
def test_line_from_a_recorded_run_is_short_and_complete(path):
    detail = json.load(open(path))
    line = bench.compact_line(detail)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT == 4096
    assert "\n" not in text
    for k in CONTRACT:
        assert k in line, k
    assert line["config"]["workload"] and "model" not in line["config"]
    for k in ROOFLINE:
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == pytest.approx(line["roofline"]["achieved"] / line["roofline"]["peak"], abs=2e-4)
    if "cpu_baseline" in detail:
        for k in CPU:
            assert line["cpu_baseline"].get(k) is not None, k
    assert line["value"] == detail["value"] and line["ms_per_step"] == detail["ms_per_step"]


def test_line_sheds_optional_scalars_rather_than_grow():
    detail = json.load(open(_records()[-1]))
    detail["multi_gpu_note"] = "x" * 6000                      # an optional scalar that alone would break the bound
    detail.setdefault("cold_start", {})["jit_compile_ms"] = {"k%d" % i: float(i) for i in range(400)}
    line = bench.compact_line(detail)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert "roofline" in line and "metric" in line and "multi_gpu" not in line


def test_traffic_json_default_is_the_newest_summary():
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_ntt_traffic.json")))[-1]
    assert bench._latest_traffic_json() == newest
