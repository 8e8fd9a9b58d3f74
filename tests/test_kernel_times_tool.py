"""tools/kernel_times.py --timed: the cut of a rocprofv3 kernel trace to the launches bench.py timed (the profile's average
has to be the timed launches', not every launch of the process).  CPU: a synthetic trace."""
import csv
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("kernel_times", os.path.join(ROOT, "tools", "kernel_times.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write_trace(path, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writeheader()
        for name, s, e in rows:
            w.writerow({"Kernel_Name": name, "Start_Timestamp": s, "End_Timestamp": e})


def test_cut_takes_exactly_the_timed_launches_of_the_pass(tmp_path):
    kt = _tool()
    shade = "void arctic::(anonymous namespace)::k_material<2, false, false>(arctic::ShadeParams)"
    other = "void arctic::(anonymous namespace)::k_material_vis<2, false, false>(arctic::ShadeParams)"
    rows, t = [], 1000
    for i in range(10):                      # launches 0..9 of the pass: 300 us cold ones first, then 200 us, written out of order below
        d = 300_000 if i < 4 else 200_000
        rows.append((shade, t, t + d)); t += d + 1000
        rows.append((other, t, t + 50_000)); t += 51_000      # never counted: another kernel
    _write_trace(str(tmp_path / "kt" / "host" / "1_kernel_trace.csv"), rows[::-1])
    bench = tmp_path / "bench.json"
    bench.write_text("some log line\n" + json.dumps({"ms_per_step": 0.2011, "roofline": {"timed_launches": [6, 10], "kernel_ms": 0.2}}) + "\n")
    out = kt.timed_cut(str(tmp_path / "kt"), str(bench))
    assert out["launches_in_trace"] == 10 and out["timed_launches"] == [6, 10]
    assert out["avg_us"] == 200.0 and out["min_us"] == 200.0 and out["max_us"] == 200.0
    assert out["all_launches_avg_us"] == pytest.approx((4 * 300 + 6 * 200) / 10)
    assert out["span_per_launch_us"] == pytest.approx((4 * 200 + 3 * 52) / 4, abs=0.01)   # back to back: what ms_per_step sees
    assert out["bench_ms_per_step"] == 0.2011


def test_cut_refuses_a_trace_that_lacks_the_timed_launches(tmp_path):
    kt = _tool()
    shade = "void arctic::(anonymous namespace)::k_material<1, false, false>(arctic::ShadeParams)"
    _write_trace(str(tmp_path / "kt" / "2_kernel_trace.csv"), [(shade, 0, 10), (shade, 20, 30)])
    bench = tmp_path / "bench.json"
    bench.write_text(json.dumps({"ms_per_step": 0.1, "roofline": {"timed_launches": [1, 5], "kernel_ms": 0.1}}))
    with pytest.raises(SystemExit):
        kt.timed_cut(str(tmp_path / "kt"), str(bench))
