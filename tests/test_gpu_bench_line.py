"""bench.py end to end at a small size: the ONE JSON line the driver parses carries every key of the contract (metric / value
/ unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload +
`roofline` + `cpu_baseline`), and the sharded loop's two forms (one stream; the overlapped three-stage pipeline on CU-masked
streams) both run to the end with an answer that passes bench.py's own invariants check."""
import json
import os
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]


def run_bench(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=900, env=e, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_default_shape_line_has_every_contract_key():
    d = run_bench("--docs", "400000", "--steps", "6", "--warmup", "2")
    assert d["metric"].startswith("queries/sec at top-k=100 (hybrid") and d["unit"] == "queries/s"
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["value"] > 0 and abs(d["value"] - 256 * 6 / (d["ms_per_step"] * 6e-3)) / d["value"] < 1e-3
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg and cfg["batch"] == 256 and cfg["k"] == 100 and cfg["query_sets"] == 4
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] in (8000.0, 2500.0)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert "traffic" in r and "traffic_from_profile" in r and r["launches"] >= 6 and r["avg_launch_ms"] > 0
    assert r["avg_launch_ms"] * r["launches_per_step"] <= d["ms_per_step"]          # the kernel fits into the step it is part of
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["unit"] == "queries/s" and c["cores"] >= 1 and c["sample"]


@pytest.mark.parametrize("overlap_cus", [None, "128"])
def test_sharded_loop_one_stream_and_overlapped(overlap_cus):
    env = {"RR_TAIL_OVERLAP_CUS": overlap_cus} if overlap_cus else {"RR_NO_TAIL_OVERLAP": "1"}
    d = run_bench("--docs", "400000", "--steps", "8", "--warmup", "2", "--force-payload", "--no-cpu-baseline", env=env)
    assert d["value"] > 0 and "cpu_baseline" not in d and d["roofline"]["launches"] >= 8
