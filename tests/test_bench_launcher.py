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


def test_launcher_retries_once_on_gloo_when_the_process_group_did_not_form(tmp_path, capsys):
    """a rank that exits with PG_FAILED_EXIT on the first launch: fresh children, a new port, --dist-backend gloo and the reason"""
    import bench
    worker = _stub(tmp_path, """
        import json, os, sys
        d = %r
        n = len([f for f in os.listdir(d) if f.startswith("launch_r%%s_" %% os.environ["RANK"])])
        open(os.path.join(d, "launch_r%%s_%%d.json" %% (os.environ["RANK"], n)), "w").write(json.dumps(dict(argv=sys.argv[1:], port=os.environ["MASTER_PORT"], pid=os.getpid())))
        if "--dist-backend-fallback" not in sys.argv:
            if os.environ["RANK"] == "2":
                sys.stderr.write("rank 2: the nccl process group did not form\\n")
                sys.exit(%d)
            import time
            time.sleep(120)      # the others wait in a rendezvous rank 2 never reaches
        if os.environ["RANK"] == "0":
            i = sys.argv.index("--dist-backend-fallback")
            print(json.dumps({"dist_backend": sys.argv[sys.argv.index("--dist-backend") + 1], "dist_backend_fallback": sys.argv[i + 1]}))
    """ % (str(tmp_path), bench.PG_FAILED_EXIT))
    t0 = time.time()
    rc = bench.launch_ranks(3, ["--gpus", "3", "--dist-backend", "nccl", "--steps", "2"], worker=worker, timeout_s=100)
    assert rc == 0 and time.time() - t0 < 60
    first = [json.loads((tmp_path / ("launch_r%d_0.json" % r)).read_text()) for r in range(3)]
    second = [json.loads((tmp_path / ("launch_r%d_1.json" % r)).read_text()) for r in range(3)]
    assert all(a["argv"] == ["--gpus", "3", "--dist-backend", "nccl", "--steps", "2"] for a in first)
    for a in second:
        assert a["argv"][:4] == ["--gpus", "3", "--steps", "2"] and a["argv"][4:6] == ["--dist-backend", "gloo"] and a["argv"][6] == "--dist-backend-fallback"
        assert a["argv"].count("--dist-backend") == 1
    assert {a["pid"] for a in first}.isdisjoint({a["pid"] for a in second})          # fresh children
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(lines) == 1                                                          # only the second launch's line is relayed
    line = json.loads(lines[0])
    assert line["dist_backend"] == "gloo" and "did not form" in line["dist_backend_fallback"]


def test_launcher_retries_only_once(tmp_path):
    import bench
    worker = _stub(tmp_path, """
        import os, sys
        open(os.path.join(%r, "seen_%%s_%%d" %% (os.environ["RANK"], os.getpid())), "w").write("x")
        sys.exit(%d)
    """ % (str(tmp_path), bench.PG_FAILED_EXIT))
    assert bench.launch_ranks(2, ["--gpus", "2"], worker=worker, timeout_s=60) == bench.PG_FAILED_EXIT
    assert 2 <= len([f for f in os.listdir(tmp_path) if f.startswith("seen_")]) <= 4   # two launches (a rank may be stopped before it writes)


def test_an_ordinary_failure_is_not_retried(tmp_path):
    import bench
    worker = _stub(tmp_path, """
        import os, sys
        open(os.path.join(%r, "seen_%%s_%%d" %% (os.environ["RANK"], os.getpid())), "w").write("x")
        sys.exit(1)
    """ % str(tmp_path))
    assert bench.launch_ranks(1, ["--gpus", "1"], worker=worker, timeout_s=60) == 1
    assert len([f for f in os.listdir(tmp_path) if f.startswith("seen_")]) == 1


def test_preflight_prints_one_json_line_and_fails_without_gpus():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this check is for boxes without a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preflight", "--gpus", "8"], capture_output=True, text=True, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 1 and len(lines) == 1
    d = json.loads(lines[0])
    assert d["preflight"] and d["gpus_asked"] == 8 and not d["ok"] and not d["checks"]["devices"]["ok"]
    assert {"devices", "distinct_devices", "free_hbm", "rccl", "host_memory", "usable_cores", "library"} <= set(d["checks"])
    assert d["checks"]["library"]["ok"]


RANKS_WORKER = """
import json, os, sys
sys.path.insert(0, %r)
import torch
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
r = bench.Ranks(torch, rank, world, 0, sys.argv[1])
r.barrier()
got = r.max_over_ranks(1.0 + rank)
ids = r.all_gather_object("dev-%%d" %% rank)
r.destroy()
if rank == 0:
    print(json.dumps({"backend": r.backend, "fallback": r.fallback, "max": got, "ids": ids}))
"""


def test_the_measurement_group_agrees_on_its_backend(tmp_path, capsys):
    """bench.Ranks on two CPU processes: asked for gloo it is gloo; asked for nccl where RCCL cannot start (no GPU here) EVERY rank
    stays on gloo, the reason is recorded, and barrier / MAX all-reduce work - no rank is left alone on another backend"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this check is for boxes without a GPU")
    import bench
    p = tmp_path / "ranks_worker.py"
    p.write_text(RANKS_WORKER % ROOT)
    for want in ("gloo", "nccl"):
        rc = bench.launch_ranks(2, [want], worker=[sys.executable, str(p)], timeout_s=240)
        assert rc == 0
        line = json.loads([l for l in capsys.readouterr().out.splitlines() if l.strip()][-1])
        assert line["backend"] == "gloo" and line["max"] == 2.0 and line["ids"] == ["dev-0", "dev-1"]
        if want == "nccl":
            assert "did not form on every rank" in line["fallback"] and "rank 0" in line["fallback"] and "rank 1" in line["fallback"]
        else:
            assert line["fallback"] is None
