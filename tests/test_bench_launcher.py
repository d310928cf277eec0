"""bench.py's multi-rank launcher on CPU: `--gpus 2` must start two ranks through
torch.distributed.run and relay rank 0's JSON line (gloo + a stub tracer stand in for RCCL + the
HIP path, which need GPUs); with the real backend it must refuse loudly when fewer than N devices
are visible instead of silently running one GPU."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(*args, timeout=240):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.timeout(300)
def test_gpus2_launches_two_ranks_strong_scaling_c5ii():
    p = run_bench("--gpus", "2", "--backend", "gloo", "--stub", "bench_stub:make", "--workload", "c5ii",
                  "--total-rays", "10001", "--subdiv", "2", "--steps", "3", "--warmup", "1", "--min-warmup-ms", "0", "--dst-share", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["steps"] == 3
    assert r["config"]["rays_total"] == 10001 and r["config"]["rays_per_gpu"] == 5001     # rank 0's shard
    assert "gathered to rank 0 inside the timed region" in r["config"]["parallelism"]
    assert r["value"] > 0 and r["metric"].startswith("Mrays/s closest-hit")
    assert "stub" in r["data"]                       # a stand-in tracer is labelled, never a measurement


@pytest.mark.timeout(300)
def test_gpus2_weak_scaling_default_workload():
    p = run_bench("--gpus", "2", "--backend", "gloo", "--stub", "bench_stub:make", "--subdiv", "2", "--res", "64",
                  "--steps", "2", "--warmup", "1", "--min-warmup-ms", "0", "--gather", "--dst-share", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak"
    assert r["config"]["rays_per_gpu"] == 64 * 64 and r["config"]["rays_total"] == 2 * 64 * 64 and r["config"]["dst_share"] is None
    # the default: rank 0 (which also finishes the other rank's rays) takes a little less than an even shard
    p = run_bench("--gpus", "2", "--backend", "gloo", "--stub", "bench_stub:make", "--subdiv", "2", "--res", "64",
                  "--steps", "2", "--warmup", "1", "--min-warmup-ms", "0")
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["scaling"] == "weak" and r["config"]["rays_total"] == 2 * 64 * 64 and r["verified"] is True
    assert 0.8 < r["config"]["dst_share"] < 1.0 and 0.4 * 2 * 64 * 64 < r["config"]["rays_per_gpu"] < 64 * 64
    # weighted shards of the weak-scaling stack: rank 0 takes fewer ROWS of the [2 x 64, 64] batch, same total
    p = run_bench("--gpus", "2", "--backend", "gloo", "--stub", "bench_stub:make", "--subdiv", "2", "--res", "64",
                  "--steps", "3", "--warmup", "1", "--min-warmup-ms", "0", "--dst-share", "0.5")
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["scaling"] == "weak" and r["config"]["rays_total"] == 2 * 64 * 64 and r["verified"] is True
    assert r["config"]["rays_per_gpu"] == 43 * 64          # round(128 rows x 0.5 / 1.5) = 43 rows on rank 0


@pytest.mark.timeout(300)
def test_gpus2_gathers_by_default_and_strong_scaling_of_one_image():
    """N > 1: the gather to rank 0 is inside the timed region by default (packed records, double-buffered
    pipeline: bench.py's `verified` compares the last step with the first call); `--scaling strong` cuts ONE
    res x res batch into row bands; `--no-gather` leaves the results in place."""
    common = ["--gpus", "2", "--backend", "gloo", "--stub", "bench_stub:make", "--subdiv", "2", "--res", "64",
              "--steps", "3", "--warmup", "1", "--min-warmup-ms", "0"]
    p = run_bench(*common)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["scaling"] == "weak" and r["config"]["rays_total"] == 2 * 64 * 64 and r["verified"] is True
    assert "gathered to rank 0 inside the timed region (4 B/ray slot records over gloo" in r["config"]["parallelism"]      # rank 0 holds the rays
    p = run_bench(*common, "--records", "packed")
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["verified"] is True and "(12 B/ray packed records over gloo" in r["config"]["parallelism"]
    p = run_bench(*common, "--scaling", "strong")
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["scaling"] == "strong" and r["config"]["rays_total"] == 64 * 64 and 24 * 64 <= r["config"]["rays_per_gpu"] <= 32 * 64
    assert r["verified"] is True and "row bands" in r["config"]["workload"]
    p = run_bench(*common, "--no-gather")
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert "gathered" not in r["config"]["parallelism"] and r["verified"] is True


@pytest.mark.timeout(300)
def test_gpus2_preflight_names_the_exchange_mode_and_both_scalings_are_in_the_line():
    """VERDICT r04 "next" #1: the N-rank line carries the exchange mode the preflight settled on (and why), the weak AND
    the strong figure; a rung that fails -- here on purpose -- costs bandwidth, not the line."""
    common = ["--gpus", "2", "--backend", "gloo", "--stub", "bench_stub:make", "--subdiv", "2", "--res", "64",
              "--steps", "3", "--warmup", "1", "--min-warmup-ms", "0"]
    p = run_bench(*common)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["config"]["exchange_mode_used"] == "slot" and r["config"]["exchange"]["attempts"] == [{"mode": "slot", "ok": True, "reason": ""}]
    assert r["scaling"] == "weak" and r["strong_64"]["rays_total"] == 64 * 64 and r["strong_64"]["verified"] is True and r["strong_64"]["value"] > 0
    assert "parity" in r and r["parity"]["pinned_bit_exact_against_reference"] is False
    env = dict(os.environ)
    os.environ["TRIRO_PREFLIGHT_FAIL"] = "slot,packed,dense"
    try:
        p = run_bench(*common, "--scaling", "strong")
    finally:
        os.environ.clear()
        os.environ.update(env)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    ex = r["config"]["exchange"]
    assert r["config"]["exchange_mode_used"] == "padded" and [a["mode"] for a in ex["attempts"]] == ["slot", "packed", "dense", "padded"]
    assert all("TRIRO_PREFLIGHT_FAIL" in a["reason"] for a in ex["attempts"][:3]) and ex["attempts"][3]["ok"]
    assert r["verified"] is True and "exchange mode 'padded'" in r["config"]["parallelism"] and r["weak"]["verified"] is True
    # every rung fails: the run is still timed -- without the gather -- and says so
    os.environ["TRIRO_PREFLIGHT_FAIL"] = "slot,packed,dense,padded,staged"
    try:
        p = run_bench(*common)
    finally:
        os.environ.clear()
        os.environ.update(env)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["config"]["exchange_mode_used"] is None and "no exchange mode passed" in r["config"]["exchange"]["error"]
    assert "gathered" not in r["config"]["parallelism"] and r["value"] > 0


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a node with fewer than 2 GPUs")
def test_gpus2_refuses_when_fewer_devices_are_visible():
    p = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert p.returncode == 2
    assert "needs 2 visible GPUs" in p.stderr and p.stdout.strip() == ""


def test_stub_is_rejected_with_the_real_backend():
    p = run_bench("--stub", "bench_stub:make", "--steps", "1")
    assert p.returncode != 0 and "only accepted with --backend gloo" in (p.stderr + p.stdout)
