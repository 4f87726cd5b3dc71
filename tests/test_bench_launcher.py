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
    # the second attempt of --ipc-legacy auto: the environment as it is (here: without the variable); a value
    # the caller set is kept either way
    envs = bench.rank_environments(2, 29511, base={"PATH": "/bin"}, ipc_legacy="env")
    assert all("HSA_ENABLE_IPC_MODE_LEGACY" not in e and e["SCARPLET_BENCH_IPC_ATTEMPT"] == "env" for e in envs)
    envs = bench.rank_environments(2, 29511, base={"HSA_ENABLE_IPC_MODE_LEGACY": "1"}, ipc_legacy="0")
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "1" for e in envs)


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

    def fake_launch(cmd, n, port=None, timeout=None, ipc_legacy="0"):
        seen["cmd"], seen["n"] = cmd, n
        seen.setdefault("ipc", []).append(ipc_legacy)
        return 0, "chatter\n" + json.dumps({"n_gpus": n}) + "\n"
    monkeypatch.setattr(bench, "launch_ranks", fake_launch)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)

    class A:
        gpus = 3
        ipc_legacy = "auto"
    with pytest.raises(SystemExit) as e:
        bench.world_or_launch(A, argv=["--gpus", "3", "--steps", "2"])
    assert e.value.code == 0
    assert seen["n"] == 3 and seen["cmd"][1].endswith("bench.py") and seen["cmd"][2:] == ["--gpus", "3", "--steps", "2"]
    assert seen["ipc"] == ["0"]                 # the first attempt worked: no second one
    # a failing rank: the exit code is passed on and nothing pretends to be a result - after BOTH attempts
    # (HSA_ENABLE_IPC_MODE_LEGACY=0, then the environment untouched)
    tried = []
    monkeypatch.setattr(bench, "launch_ranks",
                        lambda cmd, n, port=None, timeout=None, ipc_legacy="0": (tried.append(ipc_legacy), (5, ""))[1])
    with pytest.raises(SystemExit) as e:
        bench.world_or_launch(A, argv=[])
    assert e.value.code == 5 and tried == ["0", "env"]
    # the first attempt fails, the second (environment untouched) gives the line
    tried.clear()
    monkeypatch.setattr(bench, "launch_ranks",
                        lambda cmd, n, port=None, timeout=None, ipc_legacy="0":
                        (tried.append(ipc_legacy), (1, "") if ipc_legacy == "0" else (0, json.dumps({"n_gpus": n}) + "\n"))[1])
    with pytest.raises(SystemExit) as e:
        bench.world_or_launch(A, argv=[])
    assert e.value.code == 0 and tried == ["0", "env"]
    A.ipc_legacy = "env"                        # named explicitly: one attempt
    tried.clear()
    with pytest.raises(SystemExit):
        bench.world_or_launch(A, argv=[])
    assert tried == ["env"]
    # under a launcher with the right world size: this process is a rank
    monkeypatch.setenv("RANK", "2"); monkeypatch.setenv("WORLD_SIZE", "3"); monkeypatch.setenv("LOCAL_RANK", "2")
    assert bench.world_or_launch(A, argv=[]) == (2, 3, 2)


# ---- a multi-GPU run must not come back empty: RCCL failure -> host transport, same process ------
class _StubCtx(object):
    """What timed_loop and the line need of a context."""

    def __init__(self, nranks):
        self.nranks, self.destroyed = nranks, 0

    def sync(self): pass
    def profile(self, stride): pass
    def profile_get(self): return {"k_inv_cols": (4, 8.0), "k_inv_rows": (2, 6.0)}
    def set_option(self, k, v): pass
    def forget_spectra(self): pass
    def comm_destroy(self): self.destroyed += 1
    def comm_info(self): return {"nranks": self.nranks, "rank": 0 if self.nranks else -1, "device": 0, "bus_id": "0000:01:00.0"}


class _FakeDist(object):
    class ReduceOp:
        MIN, MAX = "min", "max"

    def all_reduce(self, t, op=None): pass       # one rank stands for all
    def barrier(self): pass


class _FakeTransport(object):
    def gather(self, obj, dst): return [obj]
    def broadcast_bytes(self, payload): return payload


def _args(**kw):
    class A:
        steps, warmup, prof_stride, opt, shard, halo, method, group, partition = 2, 1, 0, "", "auto", "rccl", "fft", 0, "tiles"
        mode = "exact"
    for k, v in kw.items():
        setattr(A, k, v)
    return A


def test_rccl_failure_falls_back_to_the_host_transport(monkeypatch):
    """sc_comm_init (or the first collective) raising on a rank: every rank drops its communicator and the
    sharding runs over the host transport in the same process; the line says so, rccl.nranks is 0."""
    import types
    built, ctx_host = [], _StubCtx(0)
    shared = _StubCtx(2)

    def fake_build(sh, backend, a, rank, world, device, transport, g, Template, scales, params, angles):
        built.append((sh, backend))
        if backend == "rccl":
            if sh == "orientations":
                raise RuntimeError("sc_comm_init: ncclCommInitRank(...): unhandled system error")
            steps = []

            def step():                          # tiles: set-up fine, the FIRST collective fails (the probe step)
                steps.append(1)
                raise RuntimeError("sc_halo_exchange: ncclGroupEnd(): unhandled system error")
            return {"step": step, "after_warmup": lambda: None, "plan": None, "ctx": shared, "part": "p",
                    "extra_seconds": lambda: 0.0, "gather_seconds": lambda: None, "result": lambda: None}
        return {"step": lambda: None, "after_warmup": lambda: None, "plan": None, "ctx": ctx_host, "part": "part",
                "mode": "exact" if sh == "tiles" else "float32", "work": lambda: (5, 6335) if sh == "tiles" else (36, 805),
                "extra_seconds": lambda: 0.002, "gather_seconds": (lambda: 0.004) if sh == "tiles" else (lambda: None),
                "result": lambda: None}
    monkeypatch.setattr(bench, "build_sharding", fake_build)
    dropped = []
    monkeypatch.setattr(bench, "drop_communicator", lambda device: dropped.append(device))
    g = types.SimpleNamespace(_griddata=__import__("numpy").zeros((8, 8)))
    base_line = lambda value, ms, plan, ranks_label, prof, n_gpus: {"value": value, "ms_per_step": ms, "n_gpus": n_gpus,
                                                                    "ranks": ranks_label}
    out = bench.run_shardings(_args(), 0, 2, 0, _FakeDist(), _FakeTransport(), None, g, None, [100.0], [1.0], [0.0],
                              "scarp", 64.0, base_line)
    assert built == [("orientations", "rccl"), ("orientations", "host"), ("tiles", "rccl"), ("tiles", "host")]
    assert dropped == [0, 0]                                     # both failed RCCL attempts dropped their communicator
    both = [out, out.get("c4_tiles") or out.get("orientations")]
    assert all(b["n_gpus"] == 2 and b["rccl"]["nranks"] == 0 for b in both)
    by = {("tiles" if "gather_ms" in b else "orientations"): b for b in both}
    assert by["orientations"]["transport"].startswith("host (RCCL failed: RuntimeError: sc_comm_init")
    assert by["tiles"]["transport"].startswith("host (RCCL failed: RuntimeError: sc_halo_exchange")
    # exact mode: the top-level line is the sharding that settles its near-ties (the tiles), the orientation sharding's
    # float32 line rides along
    assert "failed" not in out and out["sharding"] == "tiles" and out["config"]["mode"] == "exact"
    assert out["orientations"]["config"]["mode"] == "float32"
    # what each rank had to do, and what that costs at the one-GPU rate: the first real run is read against it
    assert out["predicted"]["tiles_x_templates_per_rank"] == [5 * 6335] and out["predicted"]["search_ms_per_step"] > 0
    assert out["orientations"]["predicted"]["tiles_per_rank"] == [36]


def test_every_transport_failing_still_prints_a_line(monkeypatch):
    import types

    def fake_build(sh, backend, *rest):
        raise RuntimeError("%s over %s: no" % (sh, backend))
    monkeypatch.setattr(bench, "build_sharding", fake_build)
    g = types.SimpleNamespace(_griddata=__import__("numpy").zeros((8, 8)))
    out = bench.run_shardings(_args(), 0, 2, 0, _FakeDist(), _FakeTransport(), None, g, None, [100.0], [1.0], [0.0],
                              "scarp", 64.0, lambda *a_: {})
    assert out["failed"] and out["value"] is None and out["n_gpus"] == 2
    assert set(out["shardings"]) == {"orientations", "tiles"} and all("error" in v for v in out["shardings"].values())


def test_gpu_telemetry_reads_sysfs(tmp_path):
    """clock_mhz / power_w of the timed loop from <pci device>/hwmon (a made-up tree here)."""
    dev = tmp_path / "0000:c1:00.0"
    hw = dev / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (hw / "freq1_input").write_text("2105000000\n")
    (hw / "power1_average").write_text("712000000\n")
    t = bench.GpuTelemetry("0000:C1:00.0", root=str(tmp_path), period=0.01).start()
    import time
    time.sleep(0.1)
    r = t.stop()
    assert r["clock_mhz"] == 2105.0 and r["power_w"] == 712.0 and r["samples"] >= 2
    # no hwmon clock: the starred level of pp_dpm_sclk
    (hw / "freq1_input").unlink()
    (dev / "pp_dpm_sclk").write_text("0: 132Mhz\n1: 2400Mhz *\n")
    t = bench.GpuTelemetry("0000:c1:00.0", root=str(tmp_path))
    t.sample()
    assert t.stop()["clock_mhz"] == 2400.0
    # nothing readable: Nones, no exception
    r = bench.GpuTelemetry("0000:99:00.0", root=str(tmp_path)).start().stop()
    assert r["clock_mhz"] is None and r["power_w"] is None
