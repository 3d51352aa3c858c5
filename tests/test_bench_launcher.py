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
    monkeypatch.setattr(b, "launcher_plan", lambda argv, n, port: [([sys.executable, str(script)], e) for _, e in real(argv, n, port)])
    import time
    t0 = time.time()
    assert b.launch_ranks(["--gpus", "3"], 3) == 1
    assert time.time() - t0 < 30
    script.write_text("import sys\nsys.exit(0)\n")
    assert b.launch_ranks(["--gpus", "2"], 2) == 0
