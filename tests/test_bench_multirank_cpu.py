"""The multi-rank paths of bench.py (what the driver launches as `bench.py --gpus N`), rehearsed on CPU ranks: gloo, the kernel emulation as
the backend, torch.cuda stubbed (tests/bench_cpu_harness.py).  No statement about RCCL or peer mapping -- only that every rank takes the
same branches through the schedule probe, the direct-exchange check, the timed loop, the real loop and the report, and that the one JSON
line carries the contract's fields."""
import json
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = ["--tiny", "--no-graph", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--frames", "2", "--height", "64", "--width", "64"]


def run_bench(world, extra):
    port = 29000 + random.randint(0, 900)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "bench_cpu_harness.py"), "--gpus", str(world)] + TINY + extra
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line: " + repr(lines)[:500]
    return json.loads(lines[0])


def test_two_ranks_default_flags_probe_time_and_report():
    d = run_bench(2, [])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert c["global_batch"] == 2 and c["parallelism"] == "dp2" and c["ranks_seen"] == 2
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]            # whole-job samples/s = ranks x clips / step time
    # --overlap auto: both RCCL schedules probed on every rank, the direct exchange refused together (no CUDA buffer here), the fastest one timed
    assert set(c["schedules"]) == {"single", "buckets"} and all("ms_per_step" in v for v in c["schedules"].values())
    assert "error" in c["direct_allreduce_check"]
    best = min(c["schedules"], key=lambda n: c["schedules"][n]["ms_per_step"])
    assert ("per-transformer-block" in c["grad_allreduce"]) == (best == "buckets")
    assert c["allreduce_ms"] > 0 and c["allreduce_bytes"] > 0
    rl = d["real_loop"]
    assert "error" not in rl and rl["losses_finite"] and rl["steps"] == 5
    assert abs(rl["value"] - 2 * 1e3 / rl["ms_per_step"]) < 1e-6 * rl["value"]
    # the roofline leg on several ranks: every rank runs the instrumented step (its collectives need all peers), rank 0 reports
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["launches"] > 0 and r["flops_per_step"] > 0 and r["kernel_ms_per_step"] > 0
    assert r["temporal_self_attention"]["levels"]


def test_four_ranks_explicit_schedule_with_gradient_accumulation():
    d = run_bench(4, ["--overlap", "buckets", "--grad-accum", "2", "--no-real-loop", "--no-roofline"])
    c = d["config"]
    assert c["ranks_seen"] == 4 and c["parallelism"] == "dp4"
    assert c["global_batch"] == 8 and abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert c["schedules"] in (None, {}) and "per-transformer-block" in c["grad_allreduce"]      # an explicit schedule is not probed
    assert c["loss"] == c["loss"] and c["opt_steps"] >= 3
