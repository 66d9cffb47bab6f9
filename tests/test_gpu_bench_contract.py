"""The driver-facing contract of bench.py, on a small workload: stdout carries exactly ONE JSON line with the agreed
keys, `roofline` and `cpu_baseline` objects included, the roofline fraction is a fraction, and the GPU answers of the
sample equal the oracle's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--n-vectors", "60000", "--nlist", "64", "--steps", "6",
           "--warmup", "1", "--query-batches", "4", "--select-batches", "2", "--cpu-sample", "64", "--compare-host-walk", "1",
           "--train-sample", "20000", "--batch", "256", "--insert-sample", "256"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[:2000]  # libraries' chatter must not reach stdout
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 6 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["unit"] == "queries/s" and j["value"] > 0 and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert abs(j["value"] - 256 * 6 / (j["ms_per_step"] * 6 / 1e3)) / j["value"] < 0.02  # value = queries / timed seconds
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.0 <= r["list_scan"]["frac"] <= 1.0
    c = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["value"] > 0 and c["gpu_matches_oracle_on_sample"] is True
    # addVectors' path: the headline's graph is the reference's sequential build, and further inserts at its full size are
    # timed on the GPU beside the CPU oracle, with identical adjacency lists
    assert j["config"]["hnsw_graph"].startswith("sequential insert")
    ins = j["insert_path"]
    for key in ("value", "unit", "graph_nodes", "sample", "cpu_value", "cpu_cores", "graph_matches_oracle_on_sample", "whole_build"):
        assert key in ins, key
    assert ins["unit"] == "inserts/s" and ins["value"] > 0 and ins["cpu_value"] > 0 and ins["cpu_cores"] == 1
    assert ins["graph_matches_oracle_on_sample"] is True and ins["device_insert"]["host_path_inserts"] == 0
