"""bench.py's multi-rank plumbing on the CPU: `--gpus N` must start N ranks itself (python -m torch.distributed.run on
127.0.0.1), shard a strong-scaled config with gfdm_amd.sharding.shard_range, and all-reduce the run statistics -- checked through
bench.py's own code path with `--selftest-launch` (gloo, no GPU, no GFDM arithmetic: the ranks checksum the synthetic input
symbols of their shard, which are keyed on the global block index)."""
import json
import os
import subprocess
import sys

import pytest


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-launch"] + list(args), env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_gpus_flag_starts_ranks_and_strong_scaled_config_is_split_invariant():
    one = run_bench("--config", "cfg4", "--batch", "1001")
    two = run_bench("--gpus", "2", "--config", "cfg4", "--batch", "1001")
    three = run_bench("--gpus", "3", "--config", "cfg4", "--batch", "1001")
    assert (one["n_gpus"], two["n_gpus"], three["n_gpus"]) == (1, 2, 3)
    assert one["scaling"] == two["scaling"] == "strong"
    assert one["blocks_per_step"] == two["blocks_per_step"] == three["blocks_per_step"] == 1001        # total work fixed
    assert two["shards"] == [[0, 501], [501, 500]] and three["shards"] == [[0, 334], [334, 334], [668, 333]]
    for a, b in zip(one["input_checksum"], two["input_checksum"]):                                     # same global blocks, any split
        assert abs(a - b) <= 1e-9 * max(1.0, abs(a))
    for a, b in zip(one["input_checksum"], three["input_checksum"]):
        assert abs(a - b) <= 1e-9 * max(1.0, abs(a))
    assert two["max_elapsed"] == pytest.approx(0.002)          # MAX over ranks of the per-rank time


@pytest.mark.timeout(600)
def test_weak_scaled_default_config_grows_with_the_ranks():
    one = run_bench("--batch", "64")
    two = run_bench("--gpus", "2", "--batch", "64")
    assert one["config"] == two["config"] == "cfg2" and two["scaling"] == "weak"
    assert (one["blocks_per_step"], two["blocks_per_step"]) == (64, 128)
    assert two["shards"][0][1] == two["shards"][1][1] == 64 and two["shards"][0][0] != two["shards"][1][0]


@pytest.mark.timeout(900)
def test_eight_ranks_shard_plan_statistics_and_placement():
    """the driver's scaling run has 8 ranks: launcher, rendezvous, shard plan, all-reduced statistics and the per-rank records
    (rank_wall_ms, rank_hosts_cpus) for the weak config and both strong ones, with every rank on a CPU slice of its own"""
    ncpu = len(os.sched_getaffinity(0))
    for cfg, batch in (("cfg2", 40), ("cfg4", 1001), ("cfg5", 203)):
        one = run_bench("--config", cfg, "--batch", str(batch))
        eight = run_bench("--gpus", "8", "--config", cfg, "--batch", str(batch))
        assert eight["n_gpus"] == 8 and len(eight["shards"]) == 8
        assert eight["rank_wall_ms"] == [1.0 * (r + 1) for r in range(8)] and eight["max_elapsed"] == pytest.approx(0.008)
        if cfg == "cfg2":
            assert eight["scaling"] == "weak" and eight["blocks_per_step"] == 8 * batch
            assert len({tuple(sh) for sh in eight["shards"]}) == 8 and all(sh[1] == batch for sh in eight["shards"])
        else:
            assert eight["scaling"] == "strong" and eight["blocks_per_step"] == batch
            edges = [sh[0] for sh in eight["shards"]] + [batch]
            assert edges[0] == 0 and all(edges[i] + eight["shards"][i][1] == edges[i + 1] for i in range(8))        # contiguous, complete
            for a, b in zip(one["input_checksum"], eight["input_checksum"]):
                assert abs(a - b) <= 1e-9 * max(1.0, abs(a))
        place = eight["rank_hosts_cpus"]
        assert len(place) == 8 and len({p["host"] for p in place}) == 1
        if ncpu >= 8:                        # disjoint slices, the launch thread on the first CPU of its slice
            assert all(p["cpus"] == ncpu // 8 and p["launch_cpu"] == p["cpu_first"] for p in place)
            assert len({p["cpu_first"] for p in place}) == 8


def test_numa_slices_follow_the_gpus():
    """bench.py moves a rank's CPU slice next to its GPU when sysfs tells where that is: on a two-socket node (GPUs 0-3 on the CPUs of socket 0,
    4-7 on socket 1, hyperthread siblings numbered behind) every rank gets an eighth of ITS socket's CPUs, disjoint from the others; an unreadable
    entry leaves the plain slices."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module_numa", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    ids = ["0000:%02x:00.0" % (0x10 + j) for j in range(8)]
    read = lambda bdf: "0-63,128-191" if int(bdf[5:7], 16) - 0x10 < 4 else "64-127,192-255"
    allowed = set(range(256))
    slices = [bench.numa_slice(ids, r, allowed, read) for r in range(8)]
    assert all(len(sl) == 32 for sl in slices) and len(set(c for sl in slices for c in sl)) == 256
    assert all(c < 64 or 128 <= c < 192 for r in range(4) for c in slices[r]) and all(64 <= c < 128 or c >= 192 for r in range(4, 8) for c in slices[r])
    assert bench.numa_slice(ids, 3, set(range(0, 256, 2)), read) == [c for c in bench.parse_cpulist("0-63,128-191") if c % 2 == 0][48:64]     # inside a restricted mask

    def missing(bdf):
        raise OSError("no such device")
    assert bench.numa_slice(ids, 0, allowed, missing) is None and bench.numa_slice(ids, 0, set(), read) is None


def test_bench_runs_under_an_external_torchrun_environment():
    """the driver's own launch: torchrun sets WORLD_SIZE, bench.py must then NOT start ranks of its own"""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "cfg5", "--batch", "10",
                        "--selftest-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["shards"] == [[0, 5], [5, 5]]


def test_roofline_traffic_comes_from_the_committed_pmc_summary():
    """bench.py's `roofline.traffic` is read from the newest profiles/rNN/pmc_hbm_traffic_summary.csv (FETCH_SIZE x 2 + WRITE_SIZE
    per launch), never hard-coded: a row exists for the default configuration's kernels at its batch, the figure is within a few
    per cent of the algorithmic bytes, an unprofiled batch gives no figure -- and neither does a summary that was measured with another
    build of the library (the rows carry gfdm_hip_build_id since round 4): then `traffic` is null and a note says why."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for kernel, bytes_per_block in (("k_row_modulate<64, 9, 2, 0>", 16 * 576), ("k_row_receive<64, 9, 2, 1, 0, 0>", 16 * 576),
                                    ("k_row_receive<64, 9, 2, 2, 1, 1>", 24 * 576)):
        tr = bench.pmc_traffic(kernel, 4096)
        assert tr["bytes"] is not None and tr["source"].startswith("profiles/r")
        assert 1.0 <= tr["bytes"] / (bytes_per_block * 4096) < 1.05
        other = bench.pmc_traffic(kernel, 4096, "0123456789abcdef")
        assert other["bytes"] is None and "build" in other["note"]
    # ... and so is `roofline.kernel_ms_rocprofv3` (profiles/rNN/kernel_alone.csv + build_id.txt): the rocprofv3 duration of the same build or null + a note
    for kernel, batch, bytes_per_block in (("k_row_modulate<64, 9, 2, 0>", 4096, 16 * 576), ("k_row_receive<64, 9, 2, 2, 1, 1>", 4096, 24 * 576),
                                           ("k_row_receive<64, 9, 2, 2, 1, 1>", 65536, 24 * 576), ("k_row_receive<128, 15, 4, 2, 0, 2>", 65536, 16 * 1920)):
        rp = bench.rocprof_kernel_ms(kernel, batch)
        assert rp["ms"] is not None and rp["source"].startswith("profiles/r")
        assert 0.3 < bytes_per_block * batch / (rp["ms"] * 1e-3) / 8e12 < 0.85             # a fraction of the 8 TB/s peak in the range this hardware gives
        other = bench.rocprof_kernel_ms(kernel, batch, "0123456789abcdef")
        assert other["ms"] is None and "build" in other["note"]
    assert bench.rocprof_kernel_ms("k_row_receive<64, 9, 2, 1, 0, 0>", 4097)["ms"] is None
    none = bench.pmc_traffic("k_row_receive<64, 9, 2, 1, 0, 0>", 4097)
    assert none["bytes"] is None and none["note"]
    assert set(bench.CONFIGS) == {"cfg2", "cfg3", "cfg4", "cfg5"} and bench.CONFIGS["cfg4"]["total"] == bench.CONFIGS["cfg5"]["total"] == 65536


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_process_group_runs_on_the_one_gpu_of_the_box():
    """torch.distributed over RCCL ("nccl") with ONE rank: process-group initialisation on the device, the barriers around the timed
    region and the all-reduces of sharding.reduce_stats all execute on the hardware that exists.  Launched the way the driver launches
    the multi-GPU bench (python -m torch.distributed.run, a child process -- nothing here has touched the GPU) and once through
    bench.py's own --force-dist; the output checksum equals the plain run's.  No scaling curve is measured by this."""
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--config", "cfg4", "--batch", "512", "--steps", "3", "--warmup", "1", "--sustained-seconds", "0", "--no-paths", "--no-cpu-baseline",
              "--ring-mib", "256"]
    bench_py = os.path.join(ROOT, "bench.py")
    runs = {"plain": [sys.executable, bench_py] + common,
            "torchrun": [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                         "--master-port", str(_free_port()), bench_py, "--gpus", "1"] + common,
            "force": [sys.executable, bench_py, "--force-dist"] + common}
    res = {}
    for name, cmd in runs.items():
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, (name, p.stderr[-3000:])
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, (name, p.stdout[-2000:])
        res[name] = json.loads(lines[0])
    assert all(r["n_gpus"] == 1 for r in res.values())
    assert res["plain"]["distributed"]["initialized"] is False
    for name in ("torchrun", "force"):
        assert res[name]["distributed"]["initialized"] is True and "RCCL" in res[name]["distributed"]["backend"]
        assert res[name]["output_checksum"] == res["plain"]["output_checksum"]
        assert res[name]["value"] > 1e6


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_whole_multi_rank_bench_path_on_one_gpu():
    """The N > 1 path of bench.py END TO END on the one GPU of the test box: `--dist-backend gloo` lets the ranks share the device, everything
    else is the path the driver runs on a multi-GPU node (launcher, rendezvous, gfdm_amd.sharding.ShardedBatch shards, the HIP kernels on
    every rank's shard, barrier-bracketed timing, all-reduced statistics).  BASELINE configs[3] is strong-scaled: the all-reduced output
    checksum of the first ring slot must not depend on the number of ranks."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = {}
    for n in (1, 2, 3):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dist-backend", "gloo", "--config", "cfg4", "--batch", "6000",
                            "--steps", "3", "--warmup", "1", "--sustained-seconds", "0", "--no-cpu-baseline", "--ring-mib", "512"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout
        res[n] = json.loads(lines[0])
    assert [res[n]["n_gpus"] for n in (1, 2, 3)] == [1, 2, 3]
    assert all(r["scaling"] == "strong" and r["config"]["blocks_per_step_all_gpus"] == 6000 for r in res.values())
    assert [res[n]["config"]["batch_per_gpu"] for n in (1, 2, 3)] == [6000, 3000, 2000]
    for n in (2, 3):
        for a, b in zip(res[1]["output_checksum"], res[n]["output_checksum"]):
            assert abs(a - b) <= 1e-9 * max(1.0, abs(a))
    assert all(r["value"] > 1e6 and r["roofline"]["frac"] > 0.05 for r in res.values())
    assert len(res[3]["rank_wall_ms"]) == 3 and max(res[3]["rank_wall_ms"]) == pytest.approx(res[3]["ms_per_step"] * 3, rel=1e-6)


@pytest.mark.gpu
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("cfg,batch", [("cfg2", 256), ("cfg4", 4000), ("cfg5", 1000)])
def test_eight_rank_bench_on_one_gpu(cfg, batch):
    """`--gpus 8` END TO END as the driver's scaling run starts it, the eight ranks sharing the one GPU of the test box through
    `--dist-backend gloo`: rc 0, one JSON line, eight per-rank records; the strong-scaled configs' all-reduced output checksum equals
    the one-rank run's (same global blocks whatever the split), the weak config processes eight batches."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = {}
    for n in (1, 8):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dist-backend", "gloo", "--config", cfg, "--batch", str(batch),
                            "--steps", "3", "--warmup", "1", "--sustained-seconds", "0", "--kernel-seconds", "0.05", "--no-cpu-baseline", "--no-paths",
                            "--ring-mib", "256"] + (["--force-dist"] if n == 1 else []), env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout
        res[n] = json.loads(lines[0])
    one, eight = res[1], res[8]
    assert eight["n_gpus"] == 8 and len(eight["rank_wall_ms"]) == 8 and len(eight["rank_hosts_cpus"]) == 8
    assert max(eight["rank_wall_ms"]) == pytest.approx(eight["ms_per_step"] * 3, rel=1e-6)          # the line's time is the slowest rank's
    assert eight["distributed"]["world_size"] == 8 and eight["value"] > 1e5
    if cfg == "cfg2":
        assert eight["scaling"] == "weak" and eight["config"]["blocks_per_step_all_gpus"] == 8 * batch and eight["config"]["batch_per_gpu"] == batch
        assert eight["output_checksum"][2] == pytest.approx(8 * one["output_checksum"][2], rel=0.02)   # eight different batches of unit-energy symbols
    else:
        assert eight["scaling"] == "strong" and eight["config"]["blocks_per_step_all_gpus"] == batch and eight["config"]["batch_per_gpu"] == batch // 8
        for a, b in zip(one["output_checksum"], eight["output_checksum"]):
            assert abs(a - b) <= 1e-9 * max(1.0, abs(a))


def _raw_bench(args, extra_env=None, timeout=300):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, capture_output=True, text=True, timeout=timeout)
    return p.returncode, [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")], p.stderr


@pytest.mark.timeout(300)
def test_more_ranks_than_visible_gpus_is_one_legible_error_line_and_a_nonzero_exit():
    """The first 8-GPU run cannot be rehearsed: if the node shows fewer GPUs than ranks under RCCL, rank 0 prints ONE JSON line with `error` and
    `devices_visible`, and the run ends with rc != 0 -- before any rank exists when bench.py starts the ranks itself, in every rank under somebody's torchrun."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box really has 8 GPUs")
    n = torch.cuda.device_count()
    rc, lines, err = _raw_bench(["--gpus", "8", "--steps", "2", "--warmup", "1"])
    assert rc == 3 and len(lines) == 1, err[-1500:]
    assert lines[0]["devices_visible"] == n and lines[0]["n_gpus"] == 8 and "8" in lines[0]["error"] and lines[0]["value"] is None
    # the driver's way: ranks started by torchrun; every rank leaves with rc != 0, only rank 0 prints
    for rank in (0, 1):
        rc, lines, err = _raw_bench(["--gpus", "8", "--steps", "2", "--warmup", "1"],
                                    {"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": "8", "LOCAL_WORLD_SIZE": "8", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
        assert rc == 3 and len(lines) == (1 if rank == 0 else 0), err[-1500:]
        assert "rank %d" % rank in err


@pytest.mark.timeout(300)
def test_a_rendezvous_that_never_completes_ends_in_an_error_line_not_in_a_hang():
    """rank 0 of a two-rank group whose rank 1 never starts: the store rendezvous gives up after --rendezvous-timeout, rank 0 prints the error line, rc = 4"""
    import time
    t0 = time.time()
    rc, lines, err = _raw_bench(["--selftest-launch", "--gpus", "2", "--rendezvous-timeout", "6"],
                                {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "LOCAL_WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    assert rc == 4 and len(lines) == 1 and "failed to form" in lines[0]["error"] and lines[0]["n_gpus"] == 2, (rc, lines, err[-1500:])
    assert time.time() - t0 < 120
