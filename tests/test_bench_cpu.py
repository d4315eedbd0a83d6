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
