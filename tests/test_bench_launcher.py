"""`python bench.py --gpus N` without torchrun's environment launches its own workers (bench.launch_workers): rehearsed on the CPU
with a stub worker (tests/launch_child.py: gloo, the GatherPipeline step loop of bench.py, one JSON line from rank 0)."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "launch_child.py")


def _run(n, argv):
    sys.path.insert(0, ROOT)
    import bench
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = bench.launch_workers(n, argv, child=CHILD, timeout=300)
    return rc, [l for l in buf.getvalue().splitlines() if l.strip()]


def test_launcher_relays_rank0_json_as_last_line():
    rc, lines = _run(2, ["--gpus", "2", "--steps", "3"])
    assert rc == 0, lines
    out = json.loads(lines[-1])
    assert out == {"world_size": 2, "steps": 3, "gathers": 3, "last_step_equals_oracle": True}
    assert sum(l.startswith("rank ") for l in lines) == 2          # the workers' ordinary output is passed through


def test_launcher_returns_nonzero_when_a_worker_fails():
    rc, lines = _run(2, ["--gpus", "2", "--steps", "1", "--fail-rank", "1"])
    assert rc != 0, lines


def test_bench_main_takes_the_launcher_path_without_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must go through launch_workers before any CUDA call."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake(n, argv, child=None, timeout=None):
        seen["n"], seen["argv"] = n, list(argv)
        return 0

    monkeypatch.setattr(bench, "launch_workers", fake)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen == {"n": 2, "argv": ["--gpus", "2", "--steps", "4", "--warmup", "1"]}
