"""bench.py's output contract on a real GPU, on a small box so that it takes seconds: exactly ONE JSON line on stdout with the
fields the driver reads (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling /
vs_baseline / dtype / data / config.workload), the `roofline` object (bound, achieved, peak, unit, frac, traffic) and the
`cpu_baseline` object (value, unit, cores, kind, sample); the N-rank path with the one RCCL rank a test box has: the three
schedules merged into config.schedules, calibration figures, the E_pol-per-cell check."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "POLAR_BENCH_LAUNCHER")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r, lines


def test_single_gpu_line_has_the_contract_fields():
    r, lines = run_bench(["--steps", "3", "--warmup", "1", "--reps", "2", "2", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, lines          # ONE line, and nothing else on stdout
    d = json.loads(lines[0])
    assert d["metric"] == "atom-steps/sec" and d["unit"] == "atom-steps/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and isinstance(d["config"]["workload"], str)
    n = d["config"]["natoms"]
    assert n == 10792 and abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.0 < rf["frac"] < 1.0
    assert "traffic" in rf and "k_field_lp" in rf["kernel"] and rf["ms_per_launch"] > 0
    assert "alone" in rf and ("frac" in rf["alone"] or "error" in rf["alone"])
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "atom-steps/s" and cb["value"] > 0 and isinstance(cb["sample"], str)
    assert d["value"] > 50 * cb["value"]      # (a sanity bound, not a claim)
    assert "extras" not in d and d["launcher_wall_s"] > 0


def test_rank_path_with_one_rccl_rank_merges_the_schedules():
    r, lines = run_bench(["--steps", "3", "--warmup", "1", "--reps", "3", "3", "3", "--no-extras"], {"POLAR_FORCE_DIST": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 1 and c["rccl_ranks"] == 1 and c["headline_schedule"] == "legacy" and c["schedule_name"] == "legacy"
    assert c["eng_pol_rel_dev_from_one_gpu"] < 1e-9
    assert set(c["schedules"]) == {"legacy", "lag1", "legacy_accel4"}
    for name, sub in c["schedules"].items():
        assert "error" not in sub, (name, sub)
        assert sub["eng_pol_rel_dev_from_one_gpu"] < 1e-9 and sub["ms_per_step"] > 0 and sub["sweeps"] > 0
        cal = sub["calibration"]
        assert cal["ms_sweep_kernels"] > 0 and cal["ms_non_sweep"] > 0 and cal["sweeps"] == sub["sweeps"] and cal["profile_intervals"] > 0
        assert cal["ms_sweep_kernels"] + cal["ms_stop_rule"] + cal["ms_accel_mixing"] <= cal["ms_solve"] * 1.02
    assert c["schedules"]["legacy_accel4"]["sweeps"] < c["schedules"]["legacy"]["sweeps"]
    assert c["schedules"]["legacy_accel4"]["calibration"]["ms_accel_mixing"] > 0 and c["schedules"]["legacy"]["calibration"]["ms_accel_mixing"] == 0
