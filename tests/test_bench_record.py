"""The driver keeps a cut-down copy of bench.py's JSON line in BENCH_rNN.json (`parsed`): of `roofline` the first 21 scalar
entries, strings cut at 120 characters, nested objects and lists dropped (observed on rounds 4 and 5, where the COMPACT and
callback-body scalars -- appended last -- were lost twice).  bench.driver_view replays that rule; these tests hold the line
to it: the north-star's kernels must be readable from what survives."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MUST_SURVIVE = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "algorithmic_bytes_per_launch",
                "kernel_ms_avg", "frac_sustained", "device_fill_GBs", "compact_all_valid_frac", "compact_holes_frac",
                "compact_holes_index_frac", "c3_32x1080p_frac", "c4_1frame_compact_us", "c4_2frames_compact_us",
                "callback_parity_ms", "callback_compact_ms", "callback_parity_valu_issue_frac")


def test_replay_of_the_driver_rule_matches_round_5_record():
    """driver_view applied to the FULL line of round 5 (its stdout tail is gone, so: to what the rule must reproduce) -- the
    recorded `parsed.roofline` of BENCH_r05.json has exactly 21 scalar keys, and its strings are cut at 120 characters."""
    rec = json.load(open(os.path.join(ROOT, "BENCH_r05.json")))["parsed"]
    assert len(rec["roofline"]) == 21
    assert len(rec["config"]["workload"]) == 120 and len(rec["cpu_baseline"]["sample"]) == 120


def test_head_keys_survive_whatever_order_they_were_added_in():
    import bench

    assert bench.ROOFLINE_HEAD[:20] == MUST_SURVIVE
    # a roofline object filled in the order main() fills it: contract fields, warm-up detail, spreads, calibration, per-rank
    # lists -- and the north-star scalars LAST, as in rounds 4 and 5
    r = {"bound": "hbm", "achieved": 6350.3, "peak": 8000.0, "unit": "GB/s", "frac": 0.79, "traffic": None, "kernel": "k",
         "algorithmic_bytes_per_launch": 1, "kernel_ms_avg": 0.39, "kernel_ms_avg_max_over_ranks": 0.39,
         "kernel_ms_avg_per_rank": [0.39], "read_component_GBs": 1.0, "frac_first_20": 0.79, "kernel_ms_first_20": 0.39,
         "warmup_launches_actual": 146, "warmup_ms_actual": 58.0, "warmup_heat_floor_ms": 60.0,
         "kernel_ms_spread": {"min": 1}, "kernel_ms_sustained_median": 0.39, "frac_sustained": 0.796,
         "device_fill_GBs": 5492.8, "device_copy_GBs": 5000.0, "calibration_what": "x" * 400}
    for i, k in enumerate(bench.ROOFLINE_HEAD[11:]):
        r[k] = 0.5 + i
    line = {"metric": "m", "value": 1.0, "config": {"workload": bench.workload_string(_Args()), "build": "abc"},
            "roofline": bench.order_roofline(r), "cpu_baseline": {"value": 1.0, "sample": "s" * 300}, "variants_1gpu": {"a": 1}}
    view = bench.driver_view(json.loads(json.dumps(line)))
    for k in MUST_SURVIVE:
        assert k in view["roofline"], k
    assert "callback_parity_clock_GHz" in view["roofline"]           # the 21st
    assert len(view["roofline"]) == 21 and "variants_1gpu" not in view
    assert view["roofline"]["frac_sustained"] == 0.796 and view["roofline"]["device_fill_GBs"] == 5492.8
    assert len(view["cpu_baseline"]["sample"]) == 120
    # nothing the run added is lost from the FULL line: the rest follows the head
    assert list(line["roofline"])[:21] == list(bench.ROOFLINE_HEAD) and "calibration_what" in line["roofline"]


class _Args:
    frames, border, mode, heat_ms, steps, warmup = 16, 40, "parity", 60.0, 200, 20


def test_workload_string_fits_the_record_and_names_the_warm_up():
    import bench

    for steps, warmup, heat, frames, border, mode in ((20, 5, 60.0, 16, 40, "parity"), (200, 20, 60.0, 16, 40, "compact"),
                                                      (100000, 1000, 1234.5, 65535, 16384, "compact")):
        a = _Args()
        a.steps, a.warmup, a.heat_ms, a.frames, a.border, a.mode = steps, warmup, heat, frames, border, mode
        w = bench.workload_string(a)
        assert len(w) <= 120, (len(w), w)
        assert "time-floored warm-up" in w and "config 4" in w and f"{steps} timed steps" in w


@pytest.mark.gpu
def test_the_emitted_line_survives_the_driver_rule():
    """bench.py as the driver runs it (fewer steps): the line it prints, passed through the driver's rule, still carries the
    COMPACT and callback-body scalars, a workload string that names the time-floored warm-up, and a live shader clock."""
    import bench

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu", "--no-host-path"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    view = bench.driver_view(json.loads(lines[0]))
    r = view["roofline"]
    for k in MUST_SURVIVE:
        assert k in r, (k, list(r))
    for k in MUST_SURVIVE[11:19]:
        assert isinstance(r[k], (int, float)) and r[k] > 0, (k, r[k])
    assert 0.5 < r["callback_parity_clock_GHz"] < 3.0
    assert "time-floored warm-up" in view["config"]["workload"] and len(view["config"]["workload"]) <= 120
    assert r["frac"] < 1.0 and r["achieved"] < r["peak"]
