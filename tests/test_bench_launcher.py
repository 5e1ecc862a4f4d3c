"""bench.py's launcher and its one-line guarantee, driven with FAKE children on the CPU: a child that hangs
after its headline, a child that dies, a schedule job that hangs, a wrong E_pol -- the headline line must survive all of them."""
import io
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def fake_child(tmp_path, name, body):
    p = tmp_path / f"{name}.py"
    p.write_text("import json, sys, time, os\n" + textwrap.dedent(body))
    return [sys.executable, str(p)]


def line(ms=9.0, dev=1e-13, **cfg):
    c = {"sweeps": 38, "iterations": 38, "eng_pol_per_cell": -7.0692, "eng_pol_rel_dev_from_one_gpu": dev, "schedule": "x", "rccl_ranks": 2}
    c.update(cfg)
    return {"metric": "atom-steps/sec", "value": 1e6 / ms, "unit": "atom-steps/s", "n_gpus": 2, "ms_per_step": ms, "config": c}


HEAD = json.dumps(line(9.0))
FULL = json.dumps(line(9.0, md_leg={"ms_per_step_md": 11.0}))


def launch(monkeypatch, cmds, n=2, budget=3.0, budget_sched=3.0, schedules=("legacy", "lag1", "legacy_accel4")):
    monkeypatch.setattr(bench, "BUDGET_HEADLINE_S", budget)
    monkeypatch.setattr(bench, "BUDGET_SCHEDULE_S", budget_sched)
    out = io.StringIO()
    rc = bench.launch(n, [], probe=lambda: 8, make_cmd=lambda name, extra: cmds[name] + list(extra), schedules=list(schedules), out=out)
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    return rc, lines


def test_headline_survives_a_child_that_hangs_in_its_extras(tmp_path, monkeypatch):
    hang = fake_child(tmp_path, "hang", f"print({HEAD!r}, flush=True)\ntime.sleep(600)\n")
    ok = fake_child(tmp_path, "ok", f"print({json.dumps(line(8.0))!r}, flush=True)\n")
    t0 = time.time()
    rc, lines = launch(monkeypatch, {"legacy": hang, "lag1": ok, "legacy_accel4": ok})
    assert time.time() - t0 < 60
    assert rc == 0 and len(lines) == 1
    d = json.loads(lines[0])
    assert d["ms_per_step"] == 9.0 and d["extras"] == "timed out"
    assert d["config"]["schedules"]["legacy"]["ms_per_step"] == 9.0
    assert d["config"]["schedules"]["lag1"]["ms_per_step"] == 8.0 and d["config"]["headline_schedule"] == "legacy"


def test_the_last_complete_line_wins_and_half_a_line_does_not(tmp_path, monkeypatch):
    two = fake_child(tmp_path, "two", f"print({HEAD!r}, flush=True)\nprint({FULL!r}, flush=True)\nsys.stdout.write({FULL[:40]!r}); sys.stdout.flush()\n")
    rc, lines = launch(monkeypatch, {"legacy": two}, schedules=("legacy",))
    assert rc == 0 and len(lines) == 1
    assert json.loads(lines[0])["config"]["md_leg"]["ms_per_step_md"] == 11.0 and "extras" not in json.loads(lines[0])


def test_a_schedule_job_that_hangs_or_dies_costs_one_sub_object(tmp_path, monkeypatch):
    ok = fake_child(tmp_path, "ok", f"print({FULL!r}, flush=True)\n")
    hang = fake_child(tmp_path, "hang", "time.sleep(600)\n")
    die = fake_child(tmp_path, "die", "sys.exit(3)\n")
    rc, lines = launch(monkeypatch, {"legacy": ok, "lag1": hang, "legacy_accel4": die}, budget_sched=2.0)
    assert rc == 0 and len(lines) == 1
    sch = json.loads(lines[0])["config"]["schedules"]
    assert sch["legacy"]["ms_per_step"] == 9.0
    assert sch["lag1"] == {"error": "timed out"}
    assert "exit code 3" in sch["legacy_accel4"]["error"]


def test_a_headline_child_that_exits_3_without_a_line_gives_a_nonzero_code_and_no_line(tmp_path, monkeypatch):
    die = fake_child(tmp_path, "die", "print('some noise', flush=True)\nsys.exit(3)\n")
    rc, lines = launch(monkeypatch, {"legacy": die}, schedules=("legacy",))
    assert rc == 3 and lines == []


def test_a_headline_child_that_exits_nonzero_after_its_line_keeps_the_line_and_the_code(tmp_path, monkeypatch):
    die = fake_child(tmp_path, "die", f"print({HEAD!r}, flush=True)\nsys.exit(5)\n")
    rc, lines = launch(monkeypatch, {"legacy": die}, schedules=("legacy",))
    assert rc == 5 and len(lines) == 1 and "exit code 5" in json.loads(lines[0])["extras"]


def test_a_wrong_energy_gives_exit_code_4_with_the_line(tmp_path, monkeypatch):
    bad = fake_child(tmp_path, "bad", f"print({json.dumps(line(9.0, dev=3e-6))!r}, flush=True)\n")
    rc, lines = launch(monkeypatch, {"legacy": bad}, schedules=("legacy",))
    assert rc == 4 and len(lines) == 1


def test_n1_runs_one_direct_child_and_relays_its_last_line(tmp_path, monkeypatch):
    ok = fake_child(tmp_path, "ok", f"print({HEAD!r}, flush=True)\nprint({FULL!r}, flush=True)\n")
    rc, lines = launch(monkeypatch, {"single": ok}, n=1, schedules=("single",))
    assert rc == 0 and len(lines) == 1
    d = json.loads(lines[0])
    assert "schedules" not in d["config"] and d["config"]["md_leg"]["ms_per_step_md"] == 11.0


def test_refuses_more_ranks_than_gpus(monkeypatch, capsys):
    out = io.StringIO()
    assert bench.launch(4, [], probe=lambda: 2, out=out) == 2 and out.getvalue() == ""
    assert "needs 4 GPUs" in capsys.readouterr().err
    assert bench.launch(1, [], probe=lambda: 0, out=out) == 2 and out.getvalue() == ""


EMITTER = """
import json, os, sys, time
sys.path.insert(0, {root!r})
import bench
em = bench.Emitter(True, 1.0)
em.headline({{"metric": "m", "value": 1.0, "config": {{}}}})
em.stage = "md_leg"
{tail}
"""


@pytest.mark.parametrize("own_launcher", [False, True])
def test_emitter_prints_exactly_one_line_under_a_foreign_launcher_even_when_the_extras_hang(tmp_path, own_launcher):
    """Under the driver's own torch.distributed.run (no POLAR_BENCH_LAUNCHER) a rank holds its headline back, and a watchdog
    prints it -- marked -- when the extras outlive their budget; the process ends with exit code 0.  Under this file's launcher
    the headline goes out at once and again, marked, at expiry (the launcher keeps the last)."""
    env = {k: v for k, v in os.environ.items() if k != "POLAR_BENCH_LAUNCHER"}
    if own_launcher:
        env["POLAR_BENCH_LAUNCHER"] = "1"
    src = tmp_path / "em.py"
    src.write_text(EMITTER.format(root=ROOT, tail="time.sleep(600)"))
    t0 = time.time()
    r = subprocess.run([sys.executable, str(src)], env=env, capture_output=True, text=True, timeout=120)
    assert time.time() - t0 < 60 and r.returncode == 0
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == (2 if own_launcher else 1)
    assert "timed out in md_leg" in lines[-1]["extras"]
    # the normal way out: one final line, no mark
    src.write_text(EMITTER.format(root=ROOT, tail='em.final({"metric": "m", "value": 2.0, "config": {"md_leg": 1}})'))
    r = subprocess.run([sys.executable, str(src)], env=env, capture_output=True, text=True, timeout=120)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines[-1]["value"] == 2.0 and "extras" not in lines[-1] and len(lines) == (2 if own_launcher else 1)


def test_a_launcher_told_to_stop_ends_its_children_and_prints_what_it_has(tmp_path):
    """SIGTERM to the launcher (a driver's own timeout): the child jobs live in sessions of their own and must not be left on the
    GPUs; the record in hand is printed, marked."""
    import signal
    pidfile = tmp_path / "child.pid"
    head = fake_child(tmp_path, "head", f"print({HEAD!r}, flush=True)\n")
    hang = fake_child(tmp_path, "hang", f"open({str(pidfile)!r}, 'w').write(str(os.getpid()))\ntime.sleep(600)\n")
    drv = tmp_path / "drv.py"
    drv.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        cmds = {{"legacy": {head!r}, "lag1": {hang!r}}}
        raise SystemExit(bench.launch(2, [], probe=lambda: 8, make_cmd=lambda name, extra: cmds[name] + list(extra), schedules=["legacy", "lag1"]))
        """))
    p = subprocess.Popen([sys.executable, str(drv)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    for _ in range(200):
        if pidfile.exists() and pidfile.read_text().strip():
            break
        time.sleep(0.1)
    child = int(pidfile.read_text())
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 128 + signal.SIGTERM
    lines = [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and lines[0]["ms_per_step"] == 9.0 and "stopped by signal" in lines[0]["extras"]
    time.sleep(0.5)
    try:                           # the schedule job is gone (or a zombie nobody has reaped yet: not running either way)
        state = [ln.split()[1] for ln in open(f"/proc/{child}/status") if ln.startswith("State:")][0]
    except (FileNotFoundError, ProcessLookupError):
        state = "gone"
    assert state in ("gone", "Z", "X"), state


def test_an_explicit_schedule_runs_that_job_alone(tmp_path, monkeypatch):
    seen = []
    ok = fake_child(tmp_path, "ok", f"print({HEAD!r}, flush=True)\n")
    monkeypatch.setattr(bench, "BUDGET_HEADLINE_S", 30.0)
    out = io.StringIO()
    rc = bench.launch(2, ["--steps", "3", "--schedule", "lag1"], probe=lambda: 8,
                      make_cmd=lambda name, extra: (seen.append((name, list(extra))) or ok), out=out)
    assert rc == 0 and seen == [("lag1", ["--schedule", "lag1"])]
    assert list(json.loads(out.getvalue())["config"]["schedules"]) == ["lag1"]
