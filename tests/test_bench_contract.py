"""bench.py's output contract (one JSON line with the driver's fields, the roofline and the CPU
baseline objects) and __graft_entry__.smoke(), run the way the driver runs them."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout           # exactly ONE line on stdout
    return json.loads(lines[0])


def test_bench_refuses_to_run_without_a_gpu_or_with_a_wrong_world():
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    # a rank whose world does not match --gpus (a launcher started with the wrong --nproc-per-node)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       cwd=ROOT, env=dict(env, WORLD_SIZE="4", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True, cwd=ROOT, env=env)
        assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


def test_bench_launches_its_own_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (children of a
    torch.distributed.run child; the parent never touches the GPU). Without a GPU both ranks refuse to
    run: the parent must then exit non-zero and print no JSON line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU variant is test_bench_two_and_eight_ranks_on_one_gpu")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, env=dict(env, NLK_BENCH_ONE_GPU="1"), timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "no CPU fallback" in r.stderr          # the ranks started and said why they stopped


def test_bench_kills_hanging_ranks_and_starts_fresh_ones():
    """The launcher's time limit without a GPU: ranks that hang (NLK_STRIPS_TEST_HANG=early: before they touch anything)
    are killed with all their descendants, FRESH ranks are started with the Python strip driver - and, there being no
    GPU here, refuse to run: a non-zero exit, no JSON line, both facts on stderr, and the parent is back in time."""
    import time
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU variant is test_bench_survives_a_hanging_strip_driver")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                       capture_output=True, text=True, cwd=ROOT, timeout=400,
                       env=dict(env, NLK_BENCH_ONE_GPU="1", NLK_STRIPS_TEST_HANG="early", NLK_BENCH_LAUNCH_TIMEOUT="30"))
    assert r.returncode != 0 and time.time() - t0 < 300
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "was killed; starting fresh ranks with --strip-driver py" in r.stderr
    assert "no CPU fallback" in r.stderr          # the fresh ranks started and said why they stopped


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 8])
def test_bench_two_and_eight_ranks_on_one_gpu(n):
    """The driver's `python bench.py --gpus N` on a one-GPU box: every rank on device 0 over gloo
    (NLK_BENCH_ONE_GPU=1) runs the whole N > 1 branch - strip plan, halo exchange, mark-word all-gather,
    whole-grid replay, accumulator halos, summed transform counts, own-row assembly, phase times."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
                        "--phase-times"], capture_output=True, text=True, cwd=ROOT, env=dict(env, NLK_BENCH_ONE_GPU="1"),
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == n and d["steps"] == 2 and d["scaling"] == "strong"
    assert d["config"]["parallelism"] == f"row strips x{n}" and d["config"]["workload"].startswith("C2: 1920x1080x3")
    assert len(d["strip_phase_ms"]) == n and all("group" in ph and "exchange_acc" in ph for ph in d["strip_phase_ms"])
    assert "cpu_baseline" not in d                 # timed at N = 1 only
    # the single-GPU line's parity fields (a sample may flip at the reference's absolute aggr > 1e-6
    # threshold in the default atomic-order mode, so max-abs is reported, the PSNR bar asserted)
    assert abs(d["psnr_delta_db"]) <= 0.02 and d["max_abs_vs_cpu"] >= 0
    assert 0 < d["roofline"]["frac"] < 1
    # the run checked itself before timing: every rank's own rows of a strip step against the whole-frame call
    assert 0 <= d["strip_selfcheck_max_abs"] <= 2e-3 and d["strip_selfcheck"]["threshold_pixels_excused"] >= 0
    assert d["launch"]["strip_driver_requested"] == "c" and "fallback" not in d["launch"]


@pytest.mark.gpu
def test_bench_survives_a_hanging_strip_driver():
    """The first real `--gpus N` run is also the first execution of the C strip driver's neighbour send / recv
    between two devices: if a step hangs, the run still ends with a line. NLK_STRIPS_TEST_HANG=1
    makes every rank that would use the C driver sleep for ever instead. Here: the ranks' OWN watchdog (what protects a
    run started by somebody else's launcher) ends them with status 5 after its fuse, and the launcher starts FRESH ranks
    on the Python strip driver and says so in the line. (The launcher's own time limit - kill the ranks and every
    descendant, start fresh ones - is exercised without a GPU: test_bench_kills_hanging_ranks_and_starts_fresh_ones.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu"],
                       capture_output=True, text=True, cwd=ROOT, timeout=500,
                       env=dict(env, NLK_BENCH_ONE_GPU="1", NLK_STRIPS_TEST_HANG="1", NLK_BENCH_C_TRIAL_TIMEOUT="15"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["launch"]["fallback"] == "py" and "watchdog" in d["launch"]["reason"]
    assert d["strip_step"]["driver"].startswith("Python") and 0 <= d["strip_selfcheck_max_abs"] <= 2e-3
    assert "status 5" in r.stderr and "starting fresh ranks with --strip-driver py" in r.stderr


def test_preflight_reports_a_hung_and_a_failed_trial(monkeypatch):
    """run_preflight (what protects ranks started by somebody else's launcher): the trial child that hangs is killed
    with its descendants at the time limit, the one that fails is reported with its status - neither touches this
    process. Without a GPU: the hang is NLK_STRIPS_TEST_HANG=early, the failure is the child's "needs a GPU"."""
    import importlib.util
    import time
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU variant is test_ranks_under_a_foreign_launcher_try_the_c_driver_in_a_child_first")
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    for k, v in (("WORLD_SIZE", "2"), ("RANK", "0"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29431"),
                 ("NLK_BENCH_ONE_GPU", "1"), ("NLK_BENCH_PREFLIGHT_TIMEOUT", "6")):
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("NLK_STRIPS_TEST_HANG", "early")
    t0 = time.time()
    r = bench.run_preflight(0)
    assert r["ok"] is False and "did not finish within 6 s and was killed" in r["why"] and time.time() - t0 < 60
    monkeypatch.delenv("NLK_STRIPS_TEST_HANG")
    monkeypatch.setenv("NLK_BENCH_PREFLIGHT_TIMEOUT", "200")
    r = bench.run_preflight(0)
    assert r["ok"] is False and "ended with status" in r["why"]


@pytest.mark.gpu
@pytest.mark.parametrize("hang", [False, True])
def test_ranks_under_a_foreign_launcher_try_the_c_driver_in_a_child_first(hang):
    """The driver starts N > 1 as `python -m torch.distributed.run ... bench.py --gpus N`: no parent of ours can time the
    ranks out. Every rank therefore tries the C strip driver's first contact in a child process before it touches its
    GPU (run_preflight); a child that hangs (NLK_STRIPS_TEST_HANG=1: ended by its own watchdog with status 5) sends
    EVERY rank to the Python driver, and the run still ends with its line. On this one-GPU box the ranks share device
    0 over gloo (NLK_BENCH_PREFLIGHT=force: the trial is normally skipped there)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(NLK_BENCH_ONE_GPU="1", NLK_BENCH_PREFLIGHT="force", NLK_BENCH_PREFLIGHT_TRIAL="10")
    if hang:
        env["NLK_STRIPS_TEST_HANG"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras"],
                       capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and 0 <= d["strip_selfcheck_max_abs"] <= 2e-3
    pf = d["preflight"]
    assert pf["ok"] is (not hang) and pf["ok_on_every_rank"] is (not hang)
    if hang:
        assert "status 5" in pf["why"] and d["strip_step"]["driver"].startswith("Python")


@pytest.mark.gpu
def test_strip_selfcheck_at_one_gpu():
    """--force-strips at N = 1 runs the same self-check (the one strip is the frame)."""
    d = _bench("--steps", "3", "--warmup", "1", "--no-cpu", "--no-extras", "--force-strips")
    assert 0 <= d["strip_selfcheck_max_abs"] <= 2e-3


@pytest.mark.gpu
def test_default_bench_line():
    d = _bench("--steps", "3", "--warmup", "1")
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Mpix/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "1080p" in d["metric"] and d["config"]["workload"].startswith("C2: 1920x1080x3")
    assert d["clock_settle_steps"] == 80      # untimed steps in front of the W warm-up steps (the clocks' ramp: bench.py)
    assert d["ms_per_step_first_20_unsettled"] > 0.9 * d["ms_per_step"]   # (reported beside, never instead)
    assert abs(d["value"] - 1920 * 1080 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * d["value"]
    ro = d["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s") and ro["peak"] > 0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3 and 0 < ro["frac"] < 1
    assert ro["traffic"] is None or ro["traffic"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    # what was timed says so itself: compiler flags, the DCT, the CPU, threads; both builds of the
    # port, one warm-up + a median each; value = the faster build
    assert cb["flags"] and cb["dct"].startswith("table") and cb["cpu"] and cb["threads"] == cb["cores"]
    assert set(cb["builds"]) == {"strict", "release"} and "-ffast-math" in cb["builds"]["release"]["flags"]
    assert cb["value"] == max(b["mpix_s"] for b in cb["builds"].values())
    assert all(b["frames_timed"] >= 1 and b["median_s"] >= b["min_s"] > 0 for b in cb["builds"].values())
    assert abs(d["psnr_delta_db"]) <= 0.02       # BASELINE.json's quality bar, on the bench frame itself
    assert d["value"] > 30 * cb["value"]          # north_star: >= 30x the CPU path on the same box
    # beside the resident call: the first frame of a sequence and the host-pointer API - never `value`
    assert d["first_frame_ms"] > d["ms_per_step"] and d["api_wall_ms"] > d["ms_per_step"]
    assert 0 < ro["valu"]["frac"] < 1 and ro["valu"]["peak"] == ro["peak"]   # one FP32 datapath: one peak


@pytest.mark.gpu
def test_strip_step_from_c_at_one_gpu():
    """--force-strips: the N > 1 step (enqueued from C over RCCL, csrc/strips.hip) with a world of one - what can be
    measured of it on a one-GPU box: its fixed cost over the whole-frame call, the host time to enqueue a step,
    the per-phase device times; with --strip-graph the step is replayed from a captured HIP graph."""
    plain = _bench("--steps", "20", "--warmup", "3", "--no-cpu")
    for extra in ((), ("--strip-graph",)):
        d = _bench("--steps", "20", "--warmup", "3", "--no-cpu", "--force-strips", "--phase-times", *extra)
        st = d["strip_step"]
        assert st["driver"].startswith("C (") and "rccl" in st["transport"]
        assert 0 < st["enqueue_us_per_step"] < 200
        assert st["hip_graph"] is (len(extra) > 0)
        assert set(d["strip_phase_ms"]) == {"exchange_prev", "match", "marks", "commit", "group", "exchange_acc", "normalize"}
        assert d["ms_per_step"] < 1.1 * plain["ms_per_step"]          # the three-phase machinery costs a few per cent
        assert 0 < d["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_flow_bench_line():
    d = _bench("--workload", "F1", "--steps", "2", "--warmup", "1")
    for k in REQUIRED:
        assert k in d, k
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1.0
    assert d["cpu_baseline"]["kind"] in ("reference", "port")
    assert d["parity_crop_480x270"]["bit_exact"] is True


@pytest.mark.gpu
def test_smoke_entry_point():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], capture_output=True, text=True,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "smoke:" in r.stdout
