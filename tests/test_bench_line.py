"""bench.py's stdout contract: the LAST (and only) stdout line is a short strict-JSON summary the driver can parse from an
8 KB tail; the full result goes to bench_detail.json.  Round 4's line was one 20 KB object and the driver recorded
`parsed: null` — these tests pin the size, the strictness and the key set on a canned full result (the builder's round-4
run, tests/golden/bench_full_result.json), and on a hostile one."""
import json
import math
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANNED = os.path.join(ROOT, "tests", "golden", "bench_full_result.json")

CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config"}
ROOFLINE_KEYS = {"bound", "achieved", "peak", "unit", "frac", "traffic"}
CPU_KEYS = {"value", "unit", "cores", "kind", "sample"}


def strict_loads(s):
    def refuse(tok):
        raise ValueError("non-strict JSON token " + tok)
    return json.loads(s, parse_constant=refuse)


def test_summary_line_is_short_strict_and_complete():
    import bench
    full = json.load(open(CANNED))
    out = bench.summary_line(full)
    assert "\n" not in out
    assert len(out.encode()) < bench.SUMMARY_MAX_BYTES <= 4096
    s = strict_loads(out)
    assert CONTRACT_KEYS <= set(s)
    assert s["metric"] == "positive+negative triples scored/sec at k=200, eta=20; filtered ranks/sec"
    assert s["value"] == full["value"] and s["ms_per_step"] == full["ms_per_step"]
    assert s["steps"] == full["steps"] and s["warmup"] == full["warmup"] and s["n_gpus"] == 1
    assert s["config"]["workload"].startswith("C3") and "model" not in s["config"]
    assert ROOFLINE_KEYS <= set(s["roofline"]) and s["roofline"]["bound"] in ("hbm", "mfma")
    assert abs(s["roofline"]["frac"] - s["roofline"]["achieved"] / s["roofline"]["peak"]) < 1e-3
    assert CPU_KEYS <= set(s["cpu_baseline"]) and s["cpu_baseline"]["kind"] in ("port", "reference")
    assert s["cpu_baseline"]["scope"] == "forward only"
    assert s["default_optimizer"]["ms_per_step"] == full["default_optimizer"]["ms_per_step"]
    assert set(s["others_ms_per_step"]) == set(full["others"])
    assert s["eval"]["exact_fast_random"]["equal_to_exact_f32_ranks"] is True
    assert s["eval"]["bf16"]["frac"] == full["eval"]["bf16"]["roofline"]["frac"]
    assert s["detail_file"] == "bench_detail.json"
    # consistency the driver checks: value = B * (1 + eta) * steps / (ms_per_step * steps)
    c = s["config"]
    assert math.isclose(s["value"], c["global_batch"] * (1 + c["eta"]) / (s["ms_per_step"] * 1e-3), rel_tol=2e-3)


def test_summary_line_survives_non_finite_numbers_and_oversized_blocks():
    import bench
    full = json.load(open(CANNED))
    full["roofline"]["traffic"] = float("nan")
    full["eval"]["bf16"]["roofline"]["frac"] = float("inf")
    full["others"] = {"W%03d" % i: {"ms_per_step": float(i)} for i in range(600)}   # would not fit: must be dropped, not truncated
    out = bench.summary_line(full)
    assert len(out.encode()) < 4096
    s = strict_loads(out)
    assert s["roofline"]["traffic"] is None
    assert CONTRACT_KEYS <= set(s) and "roofline" in s and "cpu_baseline" in s
    assert "others_ms_per_step" not in s


def test_emit_prints_exactly_one_stdout_line_and_writes_the_detail_file(tmp_path):
    code = ("import json, sys; sys.path.insert(0, %r); import bench; bench.ROOT = %r; "
            "bench.emit(json.load(open(%r)))" % (ROOT, str(tmp_path), CANNED))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096
    assert strict_loads(lines[0])["value"] == json.load(open(CANNED))["value"]
    detail = strict_loads(open(os.path.join(str(tmp_path), "bench_detail.json")).read())
    assert "others" in detail and "stages" in detail and detail["value"] == strict_loads(lines[0])["value"]
