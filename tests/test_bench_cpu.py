"""CPU: the parts of bench.py that do not need a GPU -- argument defaults and the cpu_baseline leg (the C oracle timed on
host cores), so a broken oracle build or a changed contract field shows up in the CPU test run."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_defaults_and_cpu_baseline(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert a.gpus == 1 and a.steps >= 200 and a.warmup >= 50  # >= 50 warm-up, >= 200 timed iterations
    cb = b.cpu_baseline(0.3)
    assert set(cb) == {"value", "unit", "cores", "kind", "sample"}
    assert cb["unit"] == "obs/s" and cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0


def test_self_launch_env_wiring_and_json_relay(tmp_path, capfd):
    """`python bench.py --gpus N` without a launcher starts N ranks of itself as child processes before anything touches
    the GPU: torchrun's environment per rank, rendezvous on 127.0.0.1, rank 0's JSON line relayed on stdout (everything
    else on stderr), the ranks' failure as the exit status."""
    b = _bench()
    env = b.child_env({"PATH": "x"}, 3, 8, 29511)
    assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"], env["MASTER_PORT"]) == ("3", "3", "8", "127.0.0.1", "29511")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "x"
    assert b.pick_json_line('RCCL version 2.x\n{"a": 1}\ntrailing\n') == '{"a": 1}'
    assert b.pick_json_line("no json here") is None
    child = tmp_path / "child.py"
    child.write_text(
        "import json, os, sys\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
        "print('banner from rank', r)\n"
        "if '--fail' in sys.argv and r == 1: sys.exit(7)\n"
        "if r == 0: print(json.dumps({'n_gpus': w, 'argv': sys.argv[1:], 'x': 'y' * 200000}))\n")
    rc = b.launch_ranks(3, ["--gpus", "3", "--steps", "5"], cmd=[sys.executable, str(child)], timeout=60)
    out, err = capfd.readouterr()
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout carries exactly the one JSON line"
    import json
    got = json.loads(lines[0])
    assert got["n_gpus"] == 3 and got["argv"] == ["--gpus", "3", "--steps", "5"]
    assert "[rank 1] banner from rank 1" in err and "[rank 0] banner from rank 0" in err
    rc = b.launch_ranks(2, ["--fail"], cmd=[sys.executable, str(child)], timeout=60)
    capfd.readouterr()
    assert rc == 7


def test_gpus_flag_without_launcher_goes_through_the_self_launcher(monkeypatch):
    b = _bench()
    seen = {}
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(b, "launch_ranks", lambda world, argv, **kw: seen.update(world=world, argv=list(argv)) or 0)
    try:
        b.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen == {"world": 4, "argv": ["--gpus", "4", "--steps", "20", "--warmup", "5"]}
    assert "torch" not in b.__dict__, "the launching parent must not import torch (let alone touch the GPU)"
