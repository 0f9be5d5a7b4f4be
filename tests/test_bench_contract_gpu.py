"""bench.py prints ONE JSON line with the keys the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", "13", "--steps", "2", "--warmup", "1", "--streams-leg", "2",
                        "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "proofs/s"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and isinstance(d["dtype"], str) and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["ms_per_step"] * d["value"] - 1000.0) < 1.0
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "proofs/s" and cb["sample"]
    assert d["concurrent_streams"]["commitments_match"] is True
