"""bench.py keeps its output contract: ONE JSON line on stdout with the driver's fields, the `roofline` and `cpu_baseline`
objects and the oracle `parity` check (run on the small C0 workload so it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line_c0():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c0", "--steps", "3", "--warmup", "1",
                        "--cpu-seconds", "1"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["vs_baseline"] is None and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and j["ms_per_step"] > 0
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # "valu" is what the C3 / C4 lines state (profiles/r2_k1_valu_model.json); C0 has no committed counter model and falls
    # back to compulsory HBM bytes.  Whatever the bound, it must BE one: frac <= 1.
    assert r["bound"] in ("hbm", "mfma", "valu") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.0 < r["frac"] <= 1.0
    k2 = j["roofline_k2"]
    assert k2["bound"] == "hbm" and 0.0 < k2["frac"] <= 1.0 and abs(k2["frac"] - k2["achieved"] / k2["peak"]) < 1e-9
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert j["parity"]["mismatching_pixels"] == 0 and j["parity"]["pixels"] > 0
