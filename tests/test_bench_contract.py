"""bench.py's driver contract, as far as a box without a GPU can check it: the script parses, its helpers that assemble the N > 1 part of the JSON line work on plain
inputs, a failing secondary block is reported in its place, and without a GPU every way of starting it — alone, `--gpus 2`, under torch.distributed.run — stops with
the one-line refusal instead of a traceback (there is no CPU fallback to time)."""
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import importlib
    return importlib.import_module("bench")


def test_safe_reports_a_failing_block_in_its_place():
    b = _bench()
    assert b.safe(lambda x: x + 1, 1) == 2
    r = b.safe(lambda: 1 / 0)
    assert "ZeroDivisionError" in r["error"] and r["where"]


def test_transport_report_shapes():
    b = _bench()
    fake = types.SimpleNamespace(get_backend=lambda: "nccl")
    assert b.transport_report(fake, 8, 1, None, 8, False) == {"transport": "rccl (torch.distributed, backend nccl)", "rccl_ranks_seen": 8, "ranks": 8}
    gloo = types.SimpleNamespace(get_backend=lambda: "gloo")
    r = b.transport_report(gloo, 2, 1, None, 2, True)
    assert r["rccl_ranks_seen"] is None and "gloo" in r["transport"]
    pt = types.SimpleNamespace(transport_info=lambda: {"transport": "rccl", "rccl_ranks": 4, "rccl_version": 22606, "detail": "librccl.so.1"})
    assert b.transport_report(None, 1, 4, pt, None, False)["rccl_ranks_seen"] == 4
    pt = types.SimpleNamespace(transport_info=lambda: {"transport": "peer-copy", "rccl_ranks": 0, "rccl_version": 0, "detail": "two members share GPU 0"})
    r = b.transport_report(None, 1, 2, pt, None, False)
    assert r["rccl_ranks_seen"] is None and "share GPU" in r["transport"]
    assert b.transport_report(None, 1, 1, None, None, False)["transport"].startswith("none")
    json.dumps(r)


def test_without_a_gpu_every_start_refuses_in_one_line():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this box has a GPU: the GPU suite runs the real thing (tests/test_gpu_bench.py)")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for cmd in ([sys.executable, "bench.py", "--steps", "2", "--warmup", "1"], [sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"],
                [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29611", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"]):
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode != 0
        assert "bench.py needs a GPU" in (r.stderr + r.stdout), (cmd, r.stderr[-800:])
        assert "Traceback" not in r.stderr.split("bench.py needs a GPU")[0][-2000:], r.stderr[-1500:]
