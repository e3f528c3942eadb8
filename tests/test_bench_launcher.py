"""bench.py --gpus N without a launcher around it: the process must start N ranks itself (fresh children, before any GPU
call), relay rank 0's line and fail when a rank fails.  CPU only: the ranks here are stub workers."""
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub(tmp_path, body):
    p = tmp_path / "stub_worker.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_launcher_starts_n_children_with_the_rendezvous_environment(tmp_path, capsys):
    import bench
    worker = _stub(tmp_path, """
        import json, os, sys
        env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "AFX_BENCH_LAUNCHED")}
        open(os.path.join(%r, "rank_%%s.json" %% env["RANK"]), "w").write(json.dumps(dict(env, argv=sys.argv[1:], pid=os.getpid())))
        print(json.dumps({"from_rank": env["RANK"], "n_gpus": int(env["WORLD_SIZE"])}))
    """ % str(tmp_path))
    rc = bench.launch_ranks(4, ["--gpus", "4", "--steps", "3"], worker=worker, timeout_s=60)
    assert rc == 0
    seen = [json.loads((tmp_path / ("rank_%d.json" % r)).read_text()) for r in range(4)]
    assert [s["RANK"] for s in seen] == ["0", "1", "2", "3"] and [s["LOCAL_RANK"] for s in seen] == ["0", "1", "2", "3"]
    assert all(s["WORLD_SIZE"] == "4" and s["MASTER_ADDR"] == "127.0.0.1" and s["AFX_BENCH_LAUNCHED"] == "1" for s in seen)
    assert len({s["MASTER_PORT"] for s in seen}) == 1 and len({s["pid"] for s in seen}) == 4 and os.getpid() not in {s["pid"] for s in seen}
    assert all(s["argv"] == ["--gpus", "4", "--steps", "3"] for s in seen)
    # only rank 0's line reaches stdout
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"from_rank": "0", "n_gpus": 4}


def test_launcher_fails_when_a_rank_fails_and_stops_the_others(tmp_path):
    import bench
    worker = _stub(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        time.sleep(120)     # a rank stuck in a rendezvous its peer never reaches
    """)
    t0 = time.time()
    rc = bench.launch_ranks(3, [], worker=worker, timeout_s=100)
    assert rc == 3 and time.time() - t0 < 30


def test_launcher_timeout(tmp_path):
    import bench
    worker = _stub(tmp_path, "import time\ntime.sleep(120)\n")
    t0 = time.time()
    assert bench.launch_ranks(2, [], worker=worker, timeout_s=1) == 124 and time.time() - t0 < 30


def test_bench_gpus_2_on_a_box_without_gpus_reports_the_failing_rank():
    """the real command line: main() must take the launcher branch (no WORLD_SIZE), the two ranks must start and, there
    being no GPU here, refuse to run ('no CPU fallback'), and the launcher must return non-zero"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this check is for boxes without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr and "exited with code" in r.stderr
    assert r.stdout.strip() == ""
