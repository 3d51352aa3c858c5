"""bench.py --gpus N starts N ranks itself (the flag used to be parsed and ignored).  CPU-only: the plan of
child processes is inspected, nothing is launched."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    argv = sys.argv
    sys.argv = ["bench.py"]          # importing must not look like `bench.py --gpus N`
    try:
        spec = importlib.util.spec_from_file_location("crt_bench", os.path.join(ROOT, "bench.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    finally:
        sys.argv = argv


def test_launcher_plan_for_eight_gpus():
    b = _bench()
    argv = ["--gpus", "8", "--steps", "4", "--warmup", "1"]
    assert b.requested_gpus(argv) == 8 and b.requested_gpus(["--gpus=4"]) == 4 and b.requested_gpus(["--steps", "3"]) == 1
    plan = b.launcher_plan(argv, 8, 29511, base_env={"PATH": "/usr/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert len(plan) == 8
    for r, (cmd, env) in enumerate(plan):
        assert cmd[0] == sys.executable and cmd[-len(argv):] == argv and os.path.samefile(cmd[2], os.path.join(ROOT, "bench.py"))
        assert env["RANK"] == env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "8"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29511"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/usr/bin"
    # one port for all ranks, distinct ranks
    assert len({e["MASTER_PORT"] for _, e in plan}) == 1 and len({e["RANK"] for _, e in plan}) == 8


def test_launcher_reports_a_failed_rank(tmp_path, monkeypatch):
    # the parent exits non-zero if any child does, and ends the others instead of waiting on their collectives
    b = _bench()
    script = tmp_path / "fake_rank.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(3)\ntime.sleep(60)\n")
    real = b.launcher_plan
    monkeypatch.setattr(b, "launcher_plan", lambda argv, n, port, base_env=None: [([sys.executable, str(script)], e) for _, e in real(argv, n, port, base_env)])
    import time
    t0 = time.time()
    assert b.launch_ranks(["--gpus", "3"], 3) == 1
    assert time.time() - t0 < 30
    script.write_text("import sys\nsys.exit(0)\n")
    assert b.launch_ranks(["--gpus", "2"], 2) == 0


def _json_lines(text):
    import json
    out = []
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{"):
            out.append(json.loads(ln))
    return out


def test_launcher_prints_a_diagnostic_line_for_a_failed_rank_and_for_a_hang(tmp_path, monkeypatch, capfd):
    # a multi-GPU run that dies must leave ONE JSON line with "error", the rank and the phase it was in
    b = _bench()
    script = tmp_path / "fake_rank.py"
    script.write_text(
        "import os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "sys.argv = ['bench.py']\n"
        "import importlib.util\n"
        "spec = importlib.util.spec_from_file_location('crt_bench', %r)\n"
        "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
        "ph = m.Phase(int(os.environ['RANK']))\n"
        "ph.set('comm_init')\n"
        "if os.environ['RANK'] == '1' and os.environ.get('FAKE_FAIL') == '1':\n"
        "    sys.exit(7)\n"
        "time.sleep(60)\n" % (ROOT, os.path.join(ROOT, "bench.py")))
    real = b.launcher_plan
    monkeypatch.setattr(b, "launcher_plan", lambda argv, n, port, base_env=None: [([sys.executable, str(script)], e) for _, e in real(argv, n, port, base_env)])
    # (a) rank 1 exits with 7 while in comm_init
    monkeypatch.setenv("FAKE_FAIL", "1")
    assert b.launch_ranks(["--gpus", "2"], 2) == 1
    lines = _json_lines(capfd.readouterr().out)
    assert len(lines) == 1 and lines[0]["error"] == "rank_failed" and lines[0]["n_gpus"] == 2
    assert lines[0]["failed_ranks"] == [{"rank": 1, "exit": 7, "phase": "comm_init"}]
    # (b) nobody fails, everybody hangs: the launcher's own limit ends the run and says where every rank was
    monkeypatch.setenv("FAKE_FAIL", "0")
    monkeypatch.setenv("CRT_BENCH_LAUNCH_TIMEOUT", "6")
    assert b.launch_ranks(["--gpus", "2"], 2) == 1
    lines = _json_lines(capfd.readouterr().out)
    assert len(lines) == 1 and lines[0]["error"] == "timeout"
    assert lines[0]["phases"] == {"0": "comm_init", "1": "comm_init"}


def test_launch_timeout_defaults_sit_below_the_drivers_limit():
    b = _bench()
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'CRT_BENCH_LAUNCH_TIMEOUT", "480"' in src          # the driver gave the bench 600 s in round 2
    assert b.rank_timeout_s() <= 480.0


def test_rank_watchdog_prints_rank_and_phase_then_exits(tmp_path):
    # under torch.distributed.run there is no launcher of ours: every rank watches itself
    import subprocess
    code = (
        "import os, sys, time\n"
        "sys.argv = ['bench.py']\n"
        "import importlib.util\n"
        "spec = importlib.util.spec_from_file_location('crt_bench', %r)\n"
        "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
        "ph = m.Phase(5); m.start_watchdog(ph, 8, 1.0); ph.set('first_gather')\n"
        "time.sleep(30)\n" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 3
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["error"] == "timeout" and lines[0]["rank"] == 5 and lines[0]["phase"] == "first_gather" and lines[0]["n_gpus"] == 8


def test_rank_watchdog_prints_the_finished_line_when_only_the_gather_hangs(tmp_path):
    # the image gather is untimed and comes after the measurement: a communicator that hangs must not cost the run its
    # result.  Rank 0 hands its finished line to the watchdog before it enters the gather.
    import subprocess
    code = (
        "import os, sys, time\n"
        "sys.argv = ['bench.py']\n"
        "import importlib.util\n"
        "spec = importlib.util.spec_from_file_location('crt_bench', %r)\n"
        "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
        "ph = m.Phase(0); m.start_watchdog(ph, 8, 300.0)\n"      # the whole-run limit is far away: the gather limit fires
        "ph.pending = {'metric': 'x', 'value': 123.0, 'n_gpus': 8, 'config': {'gather': None, 'gather_ms': 1.0, 'gather_check': 'ok'}}\n"
        "ph.measured_t = time.time()\n"
        "ph.set('comm_init')\n"
        "time.sleep(30)\n" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")}
    env["CRT_BENCH_GATHER_TIMEOUT"] = "1"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["value"] == 123.0 and "error" not in lines[0]
    assert lines[0]["config"]["gather"].startswith("failed: timeout in phase comm_init") and lines[0]["config"]["gather_check"] is None


def test_other_ranks_end_quietly_with_exit_code_0_when_the_gather_hangs_after_the_measurement(tmp_path):
    # under torch.distributed.run every rank's stdout is the job's stdout and a non-zero exit fails the job: a rank that is
    # stuck in the gather AFTER the measurement prints no JSON of its own and does not fail the run
    import subprocess
    code = (
        "import os, sys, time\n"
        "sys.argv = ['bench.py']\n"
        "import importlib.util\n"
        "spec = importlib.util.spec_from_file_location('crt_bench', %r)\n"
        "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
        "ph = m.Phase(3); m.start_watchdog(ph, 8, 300.0)\n"
        "ph.measured_t = time.time()\n"
        "ph.set('first_gather')\n"
        "time.sleep(30)\n" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")}
    env["CRT_BENCH_GATHER_TIMEOUT"] = "1"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0 and _json_lines(r.stdout) == [] and "rank 3 stuck in phase first_gather" in r.stderr


def test_rank_0_reports_its_phase_when_the_launcher_terminates_it(tmp_path):
    # torch.distributed.run ends the surviving ranks with SIGTERM when one rank dies: rank 0 leaves one JSON line behind
    import signal
    import subprocess
    import time
    code = (
        "import os, sys, time\n"
        "sys.argv = ['bench.py']\n"
        "import importlib.util\n"
        "spec = importlib.util.spec_from_file_location('crt_bench', %r)\n"
        "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
        "ph = m.Phase(0); m.install_term_handler(ph, 8); ph.set('timed_steps')\n"
        "print('ready', file=sys.stderr, flush=True)\n"
        "import ctypes; ctypes.CDLL(None).sleep(30)\n" % os.path.join(ROOT, "bench.py"))      # the main thread blocked in C, as in a collective
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")}
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert p.stderr.readline().strip() == "ready"
    p.send_signal(signal.SIGTERM)
    out, _ = p.communicate(timeout=30)
    assert p.returncode == 143
    lines = _json_lines(out)
    assert len(lines) == 1 and lines[0]["error"] == "terminated" and lines[0]["phase"] == "timed_steps" and lines[0]["n_gpus"] == 8
