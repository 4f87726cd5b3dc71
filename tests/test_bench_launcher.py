"""bench.py --gpus N starts N ranks itself and never reports one world size as another
(CPU only: the children here are tiny python programs, not the benchmark)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CHILD = textwrap.dedent("""
    import json, os, sys, time
    r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    mode = sys.argv[1]
    if mode == "fail" and r == w - 1:
        sys.exit(7)
    if mode == "fail":
        time.sleep(60)                      # a rank stuck in a collective: the launcher must end it
    if r == 0:
        print("noise on stdout from a library")
        print(json.dumps({"n_gpus": w, "local": os.environ["LOCAL_RANK"], "addr": os.environ["MASTER_ADDR"],
                          "port": os.environ["MASTER_PORT"]}))
    else:
        print("rank %d says hello on its own stdout" % r)
""")


def test_rank_environments():
    envs = bench.rank_environments(4, 29511, base={"PATH": "/bin", "RANK": "9"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511"
               and e["PATH"] == "/bin" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)


def test_launcher_relays_rank_zero_and_joins_all(tmp_path):
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    rc, text = bench.launch_ranks([sys.executable, str(child), "ok"], 3)
    assert rc == 0
    line = json.loads([ln for ln in text.splitlines() if ln.strip()][-1])
    assert line["n_gpus"] == 3 and line["local"] == "0" and line["addr"] == "127.0.0.1"
    assert "rank 1 says" not in text            # other ranks' stdout is not rank 0's


def test_launcher_fails_when_a_rank_fails(tmp_path):
    import time
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    t0 = time.time()
    rc, text = bench.launch_ranks([sys.executable, str(child), "fail"], 3)
    assert rc == 7                              # the failing rank's code, not 0
    assert time.time() - t0 < 30                # the ranks left waiting were terminated, not waited for


def _run_bench(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env,
                          capture_output=True, text=True, timeout=120)


def test_world_size_mismatch_is_refused():
    """WORLD_SIZE from a launcher that disagrees with --gpus: non-zero exit, no JSON line
    (before any import of the package: no GPU needed)."""
    r = _run_bench(["--gpus", "8", "--no-cpu-baseline", "--no-verify"], {"RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0
    assert "--gpus 8 but WORLD_SIZE=1" in r.stderr
    assert "n_gpus" not in r.stdout
    r = _run_bench(["--gpus", "1"], {"RANK": "0", "WORLD_SIZE": "2"})
    assert r.returncode != 0 and "n_gpus" not in r.stdout


def test_gpus_n_without_a_launcher_starts_n_ranks(monkeypatch):
    """world_or_launch(): no RANK in the environment and --gpus 3 -> launch_ranks of this very
    script with the same arguments, rank 0's last line relayed, exit code passed on."""
    seen = {}

    def fake_launch(cmd, n, port=None, timeout=None):
        seen["cmd"], seen["n"] = cmd, n
        return 0, "chatter\n" + json.dumps({"n_gpus": n}) + "\n"
    monkeypatch.setattr(bench, "launch_ranks", fake_launch)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)

    class A:
        gpus = 3
    with pytest.raises(SystemExit) as e:
        bench.world_or_launch(A, argv=["--gpus", "3", "--steps", "2"])
    assert e.value.code == 0
    assert seen["n"] == 3 and seen["cmd"][1].endswith("bench.py") and seen["cmd"][2:] == ["--gpus", "3", "--steps", "2"]
    # a failing rank: the exit code is passed on and nothing pretends to be a result
    monkeypatch.setattr(bench, "launch_ranks", lambda cmd, n, port=None, timeout=None: (5, ""))
    with pytest.raises(SystemExit) as e:
        bench.world_or_launch(A, argv=[])
    assert e.value.code == 5
    # under a launcher with the right world size: this process is a rank
    monkeypatch.setenv("RANK", "2"); monkeypatch.setenv("WORLD_SIZE", "3"); monkeypatch.setenv("LOCAL_RANK", "2")
    assert bench.world_or_launch(A, argv=[]) == (2, 3, 2)
