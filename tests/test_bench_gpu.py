"""-m gpu: the bench contract on a reduced workload -- one JSON line with the agreed keys (the driver parses exactly this)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, "bench.py", "--gaussians", "200000", "--steps", "3", "--warmup", "1", "--vq-steps", "2",
                        "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["value"] > 0
    assert d["vs_baseline"] is None and "workload" in d["config"] and d["data"].startswith("synthetic")
    rf = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rf) and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert d["device_allocs_in_timed_region"] == 0 and d["ranks_seen"] == 1
    assert set(d["roofline_blend"]) == {"render_forward", "render_backward"}
    assert all(0 < v["frac"] < 1 for v in d["roofline_blend"].values())
    vq = d["vq"]
    assert vq["value"] > 0 and vq["steps"] == 2 and vq["final_assignment_ms"] > 0 and vq["accumulate_first_step_ms"] > 0
    assert "error" not in json.dumps(d.get("qat_loop", {})) and "error" not in json.dumps(d.get("qat_model", {}))
    assert "error" not in json.dumps(d.get("vq", {})) and "error" not in json.dumps(d.get("postvq_index_layout", {}))
