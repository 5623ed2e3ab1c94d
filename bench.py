#!/usr/bin/env python3
"""bench.py -- GFDM blocks/s on MI355X for the BASELINE.json configurations.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg5]

--config (default cfg2 = BASELINE configs[1], the configuration the headline metric is quoted on):
    cfg2  K=64  M=9  L=2 alpha=0.2   step = modulate + MF demodulate of 4096 blocks PER GPU           (weak scaling)
    cfg3  K=64  M=9  L=2 alpha=0.2   step = ZF demodulate + 2 IC iterations of 4096 blocks PER GPU    (weak; north-star path)
    cfg4  K=128 M=15 L=4 alpha=0.2   step = MF demodulate + 2 IC iterations of 65 536 blocks IN TOTAL (strong: sharded over the GPUs)
    cfg5  K=256 M=31 L=2 alpha=0.1   step = ZF demodulate of 65 536 blocks IN TOTAL                   (strong)

--gpus N > 1 without a torchrun environment: this process starts N ranks itself (python -m torch.distributed.run, one rank
per GPU, rendezvous on 127.0.0.1) BEFORE anything touches the GPU and exits with their status.  Under torchrun (WORLD_SIZE
set, as the driver launches it) it is one of the ranks.  GFDM blocks are independent, so the batch shards contiguously
(gfdm_amd.sharding.shard_range) with NO data-path collective; RCCL carries the barriers, the max-over-ranks time and an output
checksum (all-reduced, so the N = 1 and N = 8 runs of a strong-scaled config can be compared).

One STEP = one pass of the hot path over one batch that is already resident in HBM.  Steps rotate through a ring of distinct
device buffers so that consecutive steps cannot be served from the 256 MiB Infinity Cache.  W warm-up steps, then exactly K
timed steps bracketed by barrier + synchronize on both sides; the time is the MAX over ranks; `value` = all blocks all ranks
processed / that time.  (The garbage collector runs BEFORE the warm-up steps and is off until the timed steps are over: a collection between warm-up and timed
steps left the GPU idle for tens of milliseconds in front of a region that is 0.3 ms long at the driver's --steps 20; GFDM_BENCH_EARLY_GC=0 restores the old order for A/B.)

Beside the headline the JSON line carries
  roofline        achieved HBM GB/s of the step's dominant (slowest) kernel = algorithmic bytes per launch (16 B/symbol,
                  24 B/symbol with the per-block equaliser vector) / its duration, measured with HIP events on
                  the launch stream.  Four readings of the duration, all in the object:
                    kernel_ms (-> achieved, frac)   MEDIAN of event pairs around SINGLE launches (a pair costs 1-2 us itself)
                    kernel_ms_pipelined             one event pair around the back-to-back run of the timed steps' launches (a burst after idle)
                    kernel_ms_sustained             back to back for --kernel-seconds after a warm-up run: the steady state
                    kernel_ms_rocprofv3             mean kernel duration of the committed rocprofv3 --kernel-trace collection
                                                    (profiles/rNN/kernel_alone.csv) -- only when it was collected with THIS build
                                                    (gfdm_hip_build_id), else null + rocprofv3_note
                  `traffic` = HBM bytes per launch from the committed rocprofv3 PMC summary of the newest profiles/rNN/, same build rule
                  (else null + `traffic_note`); `copy_ceiling_GBps` = a plain device copy of the same byte count, single launches timed
                  the same way; `north_star` (cfg2 / cfg3 runs at N = 1) = the same four readings for the ZF + 2 IC kernel of BASELINE configs[2]
  roofline_kernels  the same figures for every kernel of the step (cfg2: modulate and demodulate)
  sustained       the headline loop again for >= 2 s (the default 200 steps are a ~4 ms burst; boxes boost for short bursts)
  single_block_host_us   one generic_work(out, in) with HOST pointers through the pybind11 drop-in class (host copy into the pinned
                  GPU-mapped buffer + kernel + wait): what an unchanged GNU Radio wrapper pays per block, beside the CPU port's time
  cpu_baseline    the plain-C oracle ("port" of the reference algorithm) on this host: pinned pthreads, one kernel object per
                  thread (oracle/gfdm_oracle_bench.c), all CPUs this process may run on; single thread beside it
  paths / large_batch   (N = 1, cfg2 / cfg3 only) every receiver variant and the modulator alone on the stream, and the same
                  kernels at 65 536 blocks per launch (steady state over >= 1 s, per-launch median, the burst after idle, rocprofv3)
  rank_wall_ms / rank_hosts_cpus   (with a process group) every rank's own time for its K steps and where it ran: each rank takes a
                  slice of the allowed CPUs by LOCAL_RANK before torch is imported and pins its launch thread to the slice's first CPU
                  (--no-pin turns that off)
  paths.host_batch_{modulate,demod_mf,zf_ic2}   the *_host entry points -- what gr-gfdm's GNU Radio wrappers call, HOST pointers --
                  at 1 / 16 / 256 / 4096 / 65 536 blocks per call: blocks/s and bytes over the PCIe link per second, on pageable
                  memory (bounced through the pinned staging sets) and on memory registered with gfdm_hip_register_host (used in
                  place); `cpu_port` = the plain-C port of the reference algorithm on the same call, one thread (how a GNU Radio block
                  runs it) and all threads this process may use
"""
import argparse
import ctypes
import ctypes.util
import csv
import gc
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a streaming copy reaches

# mode: kernels of a step; total: blocks of the whole job (strong scaling) or None (batch per GPU, weak scaling)
CONFIGS = {
    "cfg2": dict(K=64, M=9, L=2, alpha=0.2, mode="mod_demod_mf", batch=4096, total=None,
                 metric="GFDM blocks/s, K=64 M=9 mod+demod (MF), batch 4096 per GPU",
                 workload="BASELINE configs[1]: K=64 subcarriers, M=9 timeslots, RRC alpha=0.2, overlap=2, MF receiver, step = modulate + demodulate"),
    "cfg3": dict(K=64, M=9, L=2, alpha=0.2, mode="zf_ic2", batch=4096, total=None,
                 metric="GFDM blocks/s, K=64 M=9 ZF demod + IC x2, batch 4096 per GPU",
                 workload="BASELINE configs[2]: K=64, M=9, RRC alpha=0.2, overlap=2, one-tap ZF equaliser + advanced receiver with 2 IC iterations"),
    "cfg4": dict(K=128, M=15, L=4, alpha=0.2, mode="mf_ic2", batch=None, total=65536,
                 metric="GFDM blocks/s, K=128 M=15 L=4 MF demod + IC x2, 65536 blocks sharded over the GPUs",
                 workload="BASELINE configs[3]: K=128, M=15, overlap=4 (RRC alpha=0.2), MF + advanced receiver with 2 IC iterations, 65536 blocks in total"),
    "cfg5": dict(K=256, M=31, L=2, alpha=0.1, mode="zf", batch=None, total=65536,
                 metric="GFDM blocks/s, K=256 M=31 ZF demod, 65536 blocks sharded over the GPUs",
                 workload="BASELINE configs[4]: K=256, M=31, RRC alpha=0.1, overlap=2, one-tap ZF equalised demodulation, 65536 blocks in total"),
}
# per mode: (uses the equaliser vector, IC iterations, algorithmic bytes per symbol of the receiver launch, cpu driver mode)
MODES = {"mod_demod_mf": (False, 0, 16, "mod_demod"), "zf_ic2": (True, 2, 24, "demod_ic"), "mf_ic2": (False, 2, 16, "demod_ic"),
         "zf": (True, 0, 24, "demod")}
GEN_CHUNK = 8192              # blocks per chunk of the on-device input generation (bounds its temporaries)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 200; 100 for the strong-scaled cfg4 / cfg5: 20 steps are a 10 ms region that ends before the launches overlap steadily, profiles/r04/bench_steps_sweep.txt)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 20)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2")
    ap.add_argument("--batch", type=int, default=None, help="override: blocks per step and per GPU (weak configs) or in total (strong configs)")
    ap.add_argument("--ring-mib", type=int, default=2048, help="total footprint of the buffer ring per path")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the independent steps of the headline loop are pipelined over (3 and 4 sustain the same rate, 3 starts a burst 3 %% faster: profiles/r04/bench_streams_sweep.txt)")
    ap.add_argument("--large-batch", type=int, default=65536, help="blocks per launch of the extra large-batch measurement (0 = skip)")
    ap.add_argument("--sustained-seconds", type=float, default=2.0, help="length of the sustained run of the headline loop (0 = skip)")
    ap.add_argument("--kernel-seconds", type=float, default=0.5, help="length of the back-to-back run behind every kernel's steady-state duration (roofline.kernel_ms_sustained; 0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-paths", action="store_true", help="skip the per-variant measurements")
    ap.add_argument("--no-host-paths", action="store_true", help="skip the host-buffer (*_host entry points) measurements")
    ap.add_argument("--host-sizes", default="1,16,256,4096,65536", help="blocks per call of the host-buffer measurements")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the barriers / stat all-reduces even with one rank")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="wall time of each cpu_baseline leg (two or three legs)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the N > 1 run (nccl = RCCL over xGMI; gloo lets several ranks share ONE GPU, "
                         "which is how the whole N > 1 path is exercised on a single-GPU box: every rank then uses device LOCAL_RANK %% device count)")
    ap.add_argument("--rendezvous-timeout", type=float, default=300.0,
                    help="N > 1: seconds the process group may take to form (store rendezvous + first barrier) before every rank gives up with rc != 0")
    ap.add_argument("--no-pin", action="store_true", help="N > 1: do not give every rank a CPU slice of its own / pin its launch thread")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="no GPU, no GFDM compute: run launcher + rendezvous (gloo) + shard plan + synthetic-input checksum all-reduce "
                         "and print them (tests/test_bench_launcher.py)")
    a = ap.parse_args(argv)
    strong = CONFIGS[a.config]["total"] is not None
    if a.steps is None:
        a.steps = 100 if strong else 200
    if a.warmup is None:
        a.warmup = 20
    return a


# ---------------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a torchrun environment starts the N ranks as a CHILD process (never an exec, and before this
# process has imported torch, let alone touched a GPU) and exits with its status

def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def fail_line(rank, rc, error, **fields):
    """A multi-GPU run that cannot start must be legible from the driver's record: rank 0 prints ONE JSON line with `error` (no `value`), every rank leaves with
    rc != 0.  Only ever called from a process that goes on to exit -- never re-exec a process that has touched the GPU."""
    if rank == 0:
        print(json.dumps(dict({"error": error, "value": None}, **fields)), flush=True)
    sys.stderr.write("bench.py rank %d: %s\n" % (rank, error))
    sys.stderr.flush()
    sys.exit(rc)


def visible_devices():
    """GPUs this process could use, WITHOUT initialising one (torch.cuda.device_count() counts through the driver's sysfs view on this image)"""
    import torch
    return torch.cuda.device_count()


def form_group(a, cfg, rank, world, dev, ndev):
    """torch.distributed process group of the run (RCCL under "nccl"; it only ever carries barriers and statistic reductions), bounded in time:
    a rendezvous that wedges (a rank that never came up, an xGMI / RCCL bring-up that hangs) must end in rc != 0 and a readable line, not in the driver's
    kill.  The store rendezvous and the first barrier are bounded by --rendezvous-timeout, a watchdog ends a rank that is stuck inside RCCL itself."""
    import datetime
    import threading
    import torch.distributed as dist
    limit = max(5.0, a.rendezvous_timeout)

    def wedged():
        if rank == 0:
            print(json.dumps({"error": "process group did not form within %.0f s (%s, %d ranks): rank 0 gave up" % (limit + 60.0, a.dist_backend, world),
                              "value": None, "n_gpus": world, "devices_visible": ndev, "metric": cfg["metric"]}), flush=True)
        os._exit(5)
    dog = threading.Timer(limit + 60.0, wedged)
    dog.daemon = True
    dog.start()
    try:
        if a.dist_backend == "nccl" and dev is not None:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=limit))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=limit))
        dist.barrier()                                                # the communicator exists and every rank is here
    except Exception as e:                                            # noqa: BLE001 -- whatever the rendezvous raises becomes the error line
        dog.cancel()
        fail_line(rank, 4, "process group (%s, %d ranks) failed to form: %s: %s" % (a.dist_backend, world, type(e).__name__, str(e)[:400]),
                  n_gpus=world, devices_visible=ndev, metric=cfg["metric"])
    dog.cancel()


def launch_ranks(a, argv):
    if a.dist_backend == "nccl" and not a.selftest_launch:
        n = visible_devices()
        if n < a.gpus:                       # before any rank exists: one GPU per rank is the contract of the RCCL run
            fail_line(0, 3, "--gpus %d needs %d visible GPUs, this node shows %d (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES=%s / %s)"
                      % (a.gpus, a.gpus, n, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES")),
                      devices_visible=n, n_gpus=a.gpus, metric=CONFIGS[a.config]["metric"])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=dict(os.environ))          # (main() has set the RCCL / OpenMP defaults already)


# ---------------------------------------------------------------------------------------------------------------------

def shard_plan(cfg, batch_override, rank, world, sharded=None):
    """(blocks of this rank per step, first global block of this rank, total blocks per step, 'weak' | 'strong'); the split is the
    product's own (gfdm_amd.sharding.ShardedBatch.shard, one device per process here)"""
    from gfdm_amd import sharding
    shard = (lambda total: sharded.shard(total, 0)) if sharded is not None else (lambda total: sharding.shard_range(total, rank, world))
    if cfg["total"] is None:
        B = batch_override or cfg["batch"]
        start, count = shard(B * world)
        assert count == B and start == rank * B
        return B, None, B * world, "weak"
    total = batch_override or cfg["total"]
    start, count = shard(total)
    return count, start, total, "strong"


def slot_block_start(plan, rank, slot):
    """first global block index of this rank's batch in ring slot `slot`: strong -> slot * total + shard start (slot 0 of all
    ranks together is exactly blocks [0, total), whatever N is); weak -> a range of its own per rank and slot"""
    B, start, total, scaling = plan
    return slot * total + start if scaling == "strong" else (rank * 1000003 + slot) * B


def gather_ranks(obj, world):
    """[obj of rank 0, obj of rank 1, ...] on every rank (outside any timed region)"""
    import torch.distributed as dist
    if world <= 1 or not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def rank_placement(cpu_slice, launch_cpu):
    """where this rank runs: host name, CPUs of its slice (count, first, last), the CPU its launch thread is pinned to"""
    return {"host": socket.gethostname(), "cpus": len(cpu_slice) if cpu_slice else None,
            "cpu_first": cpu_slice[0] if cpu_slice else None, "cpu_last": cpu_slice[-1] if cpu_slice else None, "launch_cpu": launch_cpu}


def selftest_launch(a, cfg, rank, world, cpu_slice=None):
    """Launcher / rendezvous / shard plan / stat reduction without a GPU: the ranks generate the synthetic symbols of their
    shard of slot 0 on the CPU (integer hashing, no GFDM arithmetic) and all-reduce their checksum."""
    import torch
    import torch.distributed as dist
    from gfdm_amd import sharding, synth
    if world > 1:
        a.dist_backend = "gloo"
        form_group(a, cfg, rank, world, None, 0)
    plan = shard_plan(cfg, a.batch, rank, world)
    B, start, total, scaling = plan
    N = cfg["K"] * cfg["M"]
    sym = synth.qpsk_symbols(slot_block_start(plan, rank, 0), B, N, torch.device("cpu"))
    nblocks, chk, tmax = sharding.reduce_stats(B, sharding.output_checksum(sym), 0.001 * (rank + 1), torch.device("cpu"))
    ranges = [None] * world
    if world > 1:
        dist.all_gather_object(ranges, (slot_block_start(plan, rank, 0), B))
    else:
        ranges = [(slot_block_start(plan, rank, 0), B)]
    placement = gather_ranks(rank_placement(cpu_slice, pin_launch_thread(cpu_slice)), world)
    walls = gather_ranks(1.0 * (rank + 1), world)
    if rank == 0:
        print(json.dumps({"selftest": True, "config": a.config, "n_gpus": world, "scaling": scaling, "blocks_per_step": nblocks,
                          "shards": ranges, "input_checksum": [float(v) for v in chk], "max_elapsed": tmax,
                          "rank_wall_ms": walls, "rank_hosts_cpus": placement}))
    if world > 1:
        dist.destroy_process_group()


def raw_launcher(fn, handle, out_t, in_ts, nblocks, stream_ptr):
    """Pre-marshalled ctypes call (keeps per-launch host cost at ~1-2 us)."""
    args = [handle, ctypes.c_void_p(out_t.data_ptr())] + [ctypes.c_void_p(t.data_ptr()) if t is not None else None for t in in_ts]
    args += [ctypes.c_int64(nblocks), ctypes.c_void_p(stream_ptr)]

    def go():
        rc = fn(*args)
        if rc != 0:
            raise RuntimeError("gfdm_hip launch failed: %d" % rc)
    return go


def timed_loop(step_fns, steps, warmup, world, time_kernels=True, kernel_seconds=0.0):
    """Run warmup + `steps` timed steps.  step_fns[i] is the list of launch closures of ring slot i (each closure is bound
    to its stream).  With `time_kernels` (single-stream loops only) every kernel of the step is then replayed alone over the same
    ring slots with ONE HIP event pair on the launch stream around the back-to-back run of its `steps` launches, i.e. the
    launch-to-launch duration: the kernel itself plus the ~1.3 us dispatch gap between two dependent launches (an event pair
    around every single launch would add another ~2.6 us of event packets to each).
    Each kernel is also timed launch by launch, every launch inside its own event pair (single_launch_ms): the median of those pairs
    is the kernel's own duration.
    With `kernel_seconds` > 0 each kernel is finally run back to back for that long (sustained_launch_ms): its steady-state duration.
    Returns (wall seconds of the timed region, [pipelined mean in ms of kernel j of a step] or None, [single-launch median] or None,
    [sustained mean] or None)."""
    import torch
    import torch.distributed as dist
    nslots = len(step_fns)
    early_gc = os.environ.get("GFDM_BENCH_EARLY_GC", "1") != "0"
    if early_gc:
        gc.collect()                          # before the warm-up steps, so that the GPU does not sit idle through a collection between them and the timed region
        gc.disable()                          # no collector pause (tens of ms with the tensor rings alive) inside the timed region
    for i in range(warmup):
        for f in step_fns[i % nslots]:
            f()
    if not early_gc:
        gc.collect()
        gc.disable()
    dist_on = dist.is_available() and dist.is_initialized()      # also a one-rank group (--force-dist / torchrun --nproc-per-node 1)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    with launch_pinned():                     # N > 1: the launch loop on one CPU of this rank's slice, for the timed steps only
        t0 = time.perf_counter()
        for i in range(steps):
            for f in step_fns[(warmup + i) % nslots]:
                f()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0       # this rank's time for its K steps; the caller takes the MAX over ranks
    gc.enable()
    if dist_on:
        dist.barrier()                        # closing bracket: every rank has finished before anyone moves on
    torch.cuda.synchronize()
    if not time_kernels:
        return wall, None, None, None
    kern_ms, kern_single, kern_sus = [], [], []
    for j in range(len(step_fns[0])):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(min(warmup, 4)):
            step_fns[i % nslots][j]()
        e0.record()
        for i in range(steps):
            step_fns[(warmup + i) % nslots][j]()
        e1.record()
        torch.cuda.synchronize()
        kern_ms.append(e0.elapsed_time(e1) / steps)
        # (at least 200 pairs behind a warm-up run of 50 launches, whatever --steps is: the driver's 20-step line took 20 pairs right behind an idle
        # gap and read the clock ramp of that moment -- 15.1 us against 9.2-11.6 us by every other reading of the same run, profiles/r05/)
        for i in range(50):
            step_fns[i % nslots][j]()
        kern_single.append(single_launch_ms(lambda i: step_fns[(warmup + i) % nslots][j](), max(200, min(steps, 1000))))
        kern_sus.append(sustained_launch_ms(lambda i: step_fns[i % nslots][j](), kernel_seconds, kern_single[-1])[0] if kernel_seconds > 0 else None)
    return wall, kern_ms, kern_single, kern_sus


def single_launch_ms(launch, n):
    """median duration of `n` single launches, each inside its own HIP event pair, the pairs queued back to back on the stream (event,
    launch(i), event, event, launch(i + 1), event ...): every pair spans one kernel from the end of its predecessor to its own end.  Of
    the three ways to time a kernel with events this is the one that agrees with rocprofv3's kernel duration (same box, K=64 M=9, 4096
    blocks, profiles/r04/event_timing_vs_rocprofv3.txt: rocprofv3 median 10.4 / 15.5 us MF / ZF + 2 IC, queued pairs 11.3 / 15.2 us; a
    pair around a launch on an idle GPU reads 13.5 / 16.6 us, the back-to-back mean 9.7 / 13.4 us)."""
    import torch
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for i in range(n):
        evs[i][0].record()
        launch(i)
        evs[i][1].record()
    torch.cuda.synchronize()
    times = sorted(x.elapsed_time(y) for x, y in evs)
    return times[len(times) // 2]


def sustained_launch_ms(launch, seconds, est_ms):
    """STEADY-STATE duration of one kernel: `launch(i)` back to back on the current stream for about `seconds` after a warm-up run of a
    quarter of that, one HIP event pair around the whole run -> (mean ms per launch, launches).  The first milliseconds after the GPU has been
    idle -- a synchronize, host-side set-up -- run at another clock state: bursts of 10 launches of one kernel read up to 35 % apart on the
    same box, while this figure and the per-launch event pairs of an equally long run agree within 2 % and repeat within 1 %
    (profiles/r05/sustained_vs_burst.txt)."""
    import torch
    n = max(10, int(seconds / max(est_ms * 1e-3, 1e-7)))
    for i in range(max(5, n // 4)):
        launch(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        launch(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, n


def copy_ceiling(nbytes_moved, steps, ring_mib, dev):
    """GB/s of a plain device-to-device copy that moves the same number of bytes (half read, half written) as one launch of the
    dominant kernel, timed like it (median of single launches over a ring)."""
    import torch
    n = max(1, nbytes_moved // 2)
    nslots = max(2, min(64, (ring_mib << 20) // (2 * n)))
    src = [torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255) for _ in range(nslots)]
    dst = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(nslots)]
    for i in range(4):
        dst[i % nslots].copy_(src[i % nslots])
    ms = single_launch_ms(lambda i: dst[i % nslots].copy_(src[i % nslots]), min(steps, 200))
    return 2.0 * n / (ms * 1e-3) / 1e9


def pmc_traffic(kernel_template, batch, build_id=None):
    """HBM bytes per launch of `kernel_template` at `batch` blocks from the newest committed rocprofv3 PMC summary
    (profiles/rNN/pmc_hbm_traffic_summary.csv, rows `<run>_<batch>` per counter and kernel; FETCH_SIZE x 2 [gfx950 correction
    for this access width, calibrated on a copy kernel of the same shape -- MI355X_MICROARCH.md, HBM] + WRITE_SIZE, KiB).
    Counters cannot be read from inside this process.  A summary counts only if it was measured with the library that is loaded now:
    its rows carry the build id (gfdm_hip_build_id) of the run.  Returns {"bytes", "source"} or {"bytes": None, "note": why}."""
    note = "no committed PMC summary has a row for this kernel and batch"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "pmc_hbm_traffic_summary.csv")), reverse=True):
        fetch = write = None
        try:
            for row in csv.DictReader(open(path)):
                if row["kernel"].replace(" ", "") != kernel_template.replace(" ", "") or not row["run"].endswith("_%d" % batch):
                    continue
                if build_id is not None and row.get("build_id") != build_id:
                    note = "%s was measured with build %s, the loaded library is build %s" % (os.path.relpath(path, ROOT), row.get("build_id") or "(not recorded)", build_id)
                    continue
                if row["counter"] == "FETCH_SIZE":
                    fetch = float(row["mean_KiB"])
                elif row["counter"] == "WRITE_SIZE":
                    write = float(row["mean_KiB"])
        except (OSError, KeyError, ValueError):
            continue
        if fetch is not None and write is not None:
            return {"bytes": (2.0 * fetch + write) * 1024.0, "source": os.path.relpath(path, ROOT)}
    return {"bytes": None, "note": note}


def rocprof_kernel_ms(kernel_template, batch, build_id=None):
    """mean duration (ms) of `kernel_template` at `batch` blocks per launch from the newest committed rocprofv3 --kernel-trace collection
    (profiles/rNN/kernel_alone.csv: one kernel on the GPU at a time, rows `<K>_<M>_<L>_<path>_<batch>`), only when that collection was made
    with THIS build of the library (profiles/rNN/build_id.txt).  Returns {"ms", "source"} or {"ms": None, "note": why}."""
    note = "no committed rocprofv3 collection has a row for this kernel and batch"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "kernel_alone.csv")), reverse=True):
        try:
            their = open(os.path.join(os.path.dirname(path), "build_id.txt")).read().strip()
        except OSError:
            their = None
        try:
            for row in csv.DictReader(open(path)):
                if row["kernel"].replace(" ", "") != kernel_template.replace(" ", "") or not row["label"].endswith("_%d" % batch):
                    continue
                if build_id is not None and their != build_id:
                    note = "%s was collected with build %s, the loaded library is build %s" % (os.path.relpath(path, ROOT), their or "(not recorded)", build_id)
                    break
                return {"ms": float(row["mean_us"]) * 1e-3, "source": os.path.relpath(path, ROOT)}
        except (OSError, KeyError, ValueError):
            continue
    return {"ms": None, "note": note}


def pin_rank(local_rank, local_world):
    """Give this rank a slice of its own of the CPUs the process may run on (by LOCAL_RANK), BEFORE torch is imported so that every
    thread a library starts later inherits it: on a multi-GPU node the ranks' launch loops, JIT and copy threads then do not migrate
    over each other's cores.  Returns the slice (sorted CPU list); the launch thread itself is pinned to the slice's first CPU by
    pin_launch_thread() once the set-up is over.  No-op where the platform has no affinity calls or there are fewer CPUs than ranks."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None
    per = len(cpus) // max(1, local_world)
    if local_world <= 1 or per < 1:
        return cpus
    mine = cpus[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return cpus
    return mine


def parse_cpulist(text):
    """'0-31,128-159' -> [0, ..., 31, 128, ..., 159]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def numa_slice(pci_bus_ids, local_rank, allowed, read=None):
    """This rank's share of the CPUs next to ITS GPU: every local rank's GPU (PCI bus id, as torch reports it) is looked up in sysfs
    (/sys/bus/pci/devices/<id>/local_cpulist), the ranks whose GPUs hang off the same CPUs split those (intersected with the CPUs this
    process may use) evenly in rank order.  None when anything is missing -- the caller keeps the plain slice by LOCAL_RANK.  On a two-socket
    node the plain slices put ranks 2, 3 (and their launch loops' doorbell writes) on the other socket than GPUs 2, 3."""
    if read is None:
        read = lambda bdf: open("/sys/bus/pci/devices/%s/local_cpulist" % bdf.lower()).read()
    try:
        lists = [tuple(c for c in parse_cpulist(read(b)) if c in allowed) for b in pci_bus_ids]
    except (OSError, ValueError):
        return None
    mine = lists[local_rank]
    group = [r for r, l in enumerate(lists) if l == mine]
    per = len(mine) // len(group)
    if per < 1:
        return None
    i = group.index(local_rank)
    return list(mine[i * per:(i + 1) * per])


def set_affinity_all_threads(cpus):
    """`cpus` for EVERY existing thread of this process (/proc/self/task), not only the caller: sched_setaffinity(0, ...) moves the calling thread alone,
    and by the time the GPU's PCI location is known torch / RCCL / the HIP runtime have started threads on the old slice (round-5 advisor)."""
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    done = 0
    for tid in tids:
        try:
            os.sched_setaffinity(tid, cpus)
            done += 1
        except (AttributeError, OSError, ValueError):       # a thread that ended meanwhile
            pass
    return done


_LAUNCH_PIN = {"slice": None, "cpu": None}


def pin_launch_thread(cpu_slice):
    """Arms the launch-thread pin for the timed regions: timed_loop() narrows the CALLING thread to the first CPU of its rank's slice only between the opening and
    the closing barrier of a timed region (launch_pinned()) and gives it the whole slice back afterwards.  A thread inherits its creator's mask, so a pin left
    in place would put every thread started later from the launch thread -- lazily started HIP / ROCr helpers, torch intra-op threads, the library's copy-pool
    helpers -- on that one CPU as well (round-5 advisor); inside a timed region everything has been warmed up and nothing starts threads.
    Returns the CPU the timed regions run on (None = no pinning)."""
    if not cpu_slice:
        return None
    _LAUNCH_PIN["slice"], _LAUNCH_PIN["cpu"] = list(cpu_slice), cpu_slice[0]
    return cpu_slice[0]


class launch_pinned:
    """with launch_pinned(): the calling thread on its rank's launch CPU (see pin_launch_thread); no-op when no pin is armed"""

    def __enter__(self):
        self.active = False
        if _LAUNCH_PIN["cpu"] is not None:
            try:
                os.sched_setaffinity(0, {_LAUNCH_PIN["cpu"]})
                self.active = True
            except (AttributeError, OSError):
                pass
        return self

    def __exit__(self, *exc):
        if self.active:
            try:
                os.sched_setaffinity(0, set(_LAUNCH_PIN["slice"]))
            except (AttributeError, OSError):
                pass
        return False


def allowed_cpus():
    try:
        return sorted(os.sched_getaffinity(0))
    except AttributeError:
        return list(range(os.cpu_count() or 1))


def cgroup_cpu_quota():
    """CPUs' worth of run time this process may use per period, from the cgroup (v2 cpu.max, v1 cpu.cfs_quota_us / cfs_period_us), or
    None when unlimited / not readable.  The affinity mask alone says nothing about it: a lease can see 256 CPUs and be granted 8."""
    try:
        rel = "/"
        for line in open("/proc/self/cgroup"):
            parts = line.strip().split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                rel = parts[2] or "/"
        best = None
        d = os.path.normpath("/sys/fs/cgroup" + rel)
        while d.startswith("/sys/fs/cgroup"):                     # the tightest limit on the way up
            try:
                q, per = open(os.path.join(d, "cpu.max")).read().split()
                if q != "max":
                    v = float(q) / float(per)
                    best = v if best is None else min(best, v)
            except (OSError, ValueError):
                pass
            if d == "/sys/fs/cgroup":
                break
            d = os.path.dirname(d)
        if best is not None:
            return best
    except OSError:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / per if q > 0 and per > 0 else None
    except (OSError, ValueError):
        return None


def cpu_baseline(cfg, taps, seconds):
    """The plain-C oracle (reference algorithm, restated; FFTW/VOLK are not installed) on this config's step, driven by
    oracle/gfdm_oracle_bench.c: pinned pthreads, one kernel object each, block after block (how the reference's GNU Radio
    wrappers drive its kernels), timed from the moment every thread has finished its set-up.  Leg 1: one thread.  Leg 2: one thread
    per CPU this process may run on, capped by the cgroup's CPU quota when one is set.  Where the host still delivers far less than
    threads x single-thread rate (a lease limited by something this process cannot read), leg 3 runs as many threads as leg 2's
    speed-up suggests; the faster of legs 2 / 3 is reported and `cores` is ITS thread count."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    try:    # rebuild for this host's ISA; fall back to the shipped portable build
        path = "/tmp/libgfdm_oracle_native_%d.so" % os.getpid()
        c_oracle.build(cflags=["-O3", "-march=native"], out=path)
        lib = c_oracle.load(path)
    except Exception:
        lib = c_oracle.load()
    K, M, L = cfg["K"], cfg["M"], cfg["L"]
    use_eq, ic_iter, _, cmode = MODES[cfg["mode"]]
    cpus = allowed_cpus()
    quota = cgroup_cpu_quota()
    chunk = max(1, 16384 // (K * M) * 4)
    run = lambda T: c_oracle.bench_threads(M, K, L, taps, cmode, T, seconds, use_eq=use_eq, ic_iter=ic_iter, cpus=cpus, chunk=chunk, lib=lib)
    n1, t1 = run(1)
    single = n1 / t1
    T_all = len(cpus) if quota is None else max(1, min(len(cpus), int(quota + 0.999)))
    legs = []
    nT, tT = run(T_all)
    legs.append({"threads": T_all, "blocks": nT, "seconds": tT, "value": nT / tT})
    speedup = legs[0]["value"] / single
    if T_all > 2 and speedup < 0.5 * T_all:
        T2 = max(2, min(T_all, int(round(speedup * 1.25))))
        n2, t2 = run(T2)
        legs.append({"threads": T2, "blocks": n2, "seconds": t2, "value": n2 / t2})
    best = max(legs, key=lambda g: g["value"])
    model, phys = "", set()
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
        for c in cpus[:best["threads"]]:
            phys.add(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip())
    except OSError:
        pass
    return {"value": best["value"], "unit": "blocks/s", "cores": best["threads"], "kind": "port",
            "sample": "%s of QPSK blocks (K=%d M=%d L=%d) for %.1f s on %d pinned pthreads, one plain-C oracle kernel object per thread "
                      "(-O3 -march=native, oracle/gfdm_oracle_bench.c; set-up not timed): %d blocks; single thread: %d blocks in %.1f s"
                      % (cfg["mode"], K, M, L, best["seconds"], best["threads"], best["blocks"], n1, t1),
            "single_thread_value": single, "single_thread_us_per_block": 1e6 * t1 / n1,
            "scaling_vs_single_thread": best["value"] / single,
            # what this process was really given: CPUs in the affinity mask, CPUs' worth of quota in the cgroup (None = no limit set),
            # and the speed-up the threads delivered (effective cores = min of the three is what `cores` should be read against)
            "cpus_in_affinity_mask": len(cpus), "cgroup_cpu_quota": quota, "effective_cores": min([len(cpus), best["value"] / single] + ([quota] if quota else [])),
            "legs": legs,
            "cpu_model": model, "host_logical_cpus": os.cpu_count(), "physical_cores_used": len(phys) or None,
            # the reference's own CPU kernels need FFTW3f and VOLK; neither is installed on the boxes of this pool (probe),
            # so the only CPU figure is the plain-C restatement of the same per-block algorithm
            "reference_libs_present": {"fftw3f": ctypes.util.find_library("fftw3f") is not None, "volk": ctypes.util.find_library("volk") is not None}}


def host_batch_paths(cfg, taps, sizes, seconds=0.3, cpu_seconds=1.0, with_cpu=True):
    """The *_host entry points (HOST pointers: what gr-gfdm's GNU Radio wrappers call, lib/simple_receiver_cc_impl.cc:61-77,
    lib/advanced_receiver_sb_cc_impl.cc:86-123, lib/simple_modulator_cc_impl.cc:62-80) at `sizes` blocks per call, on pageable numpy
    memory and on the same arrays registered with gfdm_hip_register_host.  Per call size: blocks/s, algorithmic bytes over the PCIe link
    per second (16 N per block, 24 N with the equaliser vector), microseconds per call, kernel launches of the call.  Beside them the
    plain-C port of the reference algorithm on the same work: one thread (a GNU Radio block runs its kernel on one thread) and every
    thread this process may use."""
    import numpy as np
    import gfdm_amd
    K, M, L = cfg["K"], cfg["M"], cfg["L"]
    N = K * M
    qpsk = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2)
    mod = gfdm_amd.Modulator(M, K, L, taps)
    dem = gfdm_amd.Demodulator(M, K, L, np.conj(taps))
    adv = gfdm_amd.AdvancedReceiver(M, K, L, np.conj(taps), np.arange(K), 2, qpsk)
    Lb = gfdm_amd.lib()
    nmax = max(sizes)
    bits = np.random.default_rng(0x6FD1).integers(0, 2, (nmax, N, 2), dtype=np.int8)
    # page-aligned arrays: the registered legs pin them (gfdm_hip_register_host takes whole pages)
    sym = gfdm_amd.aligned_empty((nmax, N))
    sym.real = (2 * bits[:, :, 0] - 1) * np.float32(np.sqrt(0.5))
    sym.imag = (2 * bits[:, :, 1] - 1) * np.float32(np.sqrt(0.5))
    del bits
    frames = gfdm_amd.aligned_empty((nmax, N))
    mod.modulate(sym, out=frames)
    feq = gfdm_amd.aligned_empty((nmax, N))
    feq[...] = 1.0
    out = gfdm_amd.aligned_empty((nmax, N))
    vp = lambda x: None if x is None else ctypes.c_void_p(x.ctypes.data)
    calls = {
        "host_batch_modulate": (Lb.gfdm_hip_modulator_work_host, [mod._h, vp(out), vp(sym)], 16, ("mod", False, 0)),
        "host_batch_demod_mf": (Lb.gfdm_hip_receiver_demodulate_host, [dem._h, vp(out), vp(frames), None], 16, ("demod", False, 0)),
        "host_batch_zf_ic2": (Lb.gfdm_hip_advanced_receiver_work_host, [adv._h, vp(out), vp(frames), vp(feq)], 24, ("demod_ic", True, 2)),
    }

    def rate(fn, args, bps, nb):
        full = args + [ctypes.c_int64(nb)]
        for _ in range(2):
            assert fn(*full) == 0
        n, t0 = 0, time.perf_counter()
        while True:
            assert fn(*full) == 0
            n += 1
            dt = time.perf_counter() - t0
            if dt > seconds and n >= 3:
                break
        st = gfdm_amd.host_call_stats()
        # where the calling thread spent the LAST call (gfdm_hip_host_call_times): sorting operands, bounce copies, kernel launches, ticket launches, waiting
        return {"blocks_per_s": nb * n / dt, "link_GBps": bps * N * nb * n / dt / 1e9, "us_per_call": dt / n * 1e6, "launches_per_call": st["chunks"],
                "in_place_operands": bin(st["direct_mask"]).count("1"), "copy_threads": st["copy_threads"],
                "last_call_us": {k: round(v / 1e3, 2) for k, v in st["ns"].items()}}

    res = {}
    for name, (fn, args, bps, _) in calls.items():
        res[name] = {"algorithmic_link_bytes_per_block": bps * N, "pageable": {str(nb): rate(fn, args, bps, nb) for nb in sizes}}
    with gfdm_amd.registered_host(sym, frames, feq, out):
        for name, (fn, args, bps, _) in calls.items():
            res[name]["registered"] = {str(nb): rate(fn, args, bps, nb) for nb in sizes}
    mode, chunk, depth, threads, streams = gfdm_amd.get_host_pipeline()
    settings = {"link_route": mode, "chunk_bytes": chunk or "auto", "staging_sets": depth, "copy_threads": threads, "kernel_streams": streams}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import c_oracle
        try:
            path = "/tmp/libgfdm_oracle_native_%d.so" % os.getpid()
            if not os.path.exists(path):
                c_oracle.build(cflags=["-O3", "-march=native"], out=path)
            lib = c_oracle.load(path)
        except Exception:
            lib = c_oracle.load()
        cpus = allowed_cpus()
        quota = cgroup_cpu_quota()
        T_all = len(cpus) if quota is None else max(1, min(len(cpus), int(quota + 0.999)))
        chunk_c = max(1, 16384 // N * 4)
        for name, (_, _, _, (cmode, use_eq, ic)) in calls.items():
            n1, t1 = c_oracle.bench_threads(M, K, L, taps, cmode, 1, cpu_seconds, use_eq=use_eq, ic_iter=ic, cpus=cpus, chunk=chunk_c, lib=lib)
            nT, tT = c_oracle.bench_threads(M, K, L, taps, cmode, T_all, cpu_seconds, use_eq=use_eq, ic_iter=ic, cpus=cpus, chunk=chunk_c, lib=lib)
            big = str(max(sizes))
            res[name]["cpu_port"] = {"kind": "port", "single_thread_blocks_per_s": n1 / t1, "threads": T_all, "blocks_per_s": nT / tT,
                                     "gpu_pageable_over_cpu_single_thread": res[name]["pageable"][big]["blocks_per_s"] / (n1 / t1),
                                     "gpu_pageable_over_cpu_all_threads": res[name]["pageable"][big]["blocks_per_s"] / (nT / tT),
                                     "at_blocks_per_call": int(big)}
    for name in res:
        res[name]["settings"] = settings
    return res


def single_block_host(cfg, taps, reps=300):
    """One block through the literal drop-in path: gfdm_python.Demodulator.demodulate(ndarray) = receiver_kernel_cc::generic_work
    with host pointers (small calls: pinned GPU-mapped buffer, kernel, stream sync) -- what gr-gfdm's unchanged wrappers call once per block
    (lib/simple_receiver_cc_impl.cc:70-74)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "lib"))
    import gfdm_python
    K, M, L = cfg["K"], cfg["M"], cfg["L"]
    dem = gfdm_python.Demodulator(M, K, L, taps)
    x = (np.random.default_rng(0).standard_normal(K * M) + 0j).astype(np.complex64)
    for _ in range(20):
        dem.demodulate(x)
    t0 = time.perf_counter()
    for _ in range(reps):
        dem.demodulate(x)
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    argv = sys.argv[1:]
    # before torch / RCCL are imported, and also when the ranks were started by somebody else's torchrun: dmabuf IPC (RCCL needs it on
    # this host driver), one OpenMP thread per rank
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    a = parse(argv)
    cfg = CONFIGS[a.config]
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a, argv))                   # children do the work; nothing in THIS process has touched a GPU
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    try:
        allowed_cpus_at_start = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed_cpus_at_start = []
    cpu_slice = None if (a.no_pin or world == 1) else pin_rank(local % max(1, local_world), local_world)     # before torch is imported
    if a.selftest_launch:
        return selftest_launch(a, cfg, rank, world, cpu_slice)

    import numpy as np
    import torch
    import torch.distributed as dist
    ndev = visible_devices()
    need = 1 if (world == 1 or a.dist_backend == "gloo") else local_world      # RCCL: one GPU per rank of this node
    if ndev < need:
        fail_line(rank, 3, ("%d ranks on this node need %d GPUs, %d visible" % (local_world, need, ndev)) if ndev else "bench.py needs an MI355X: the HIP kernels have no CPU fallback",
                  devices_visible=ndev, n_gpus=world, ranks_on_this_node=local_world, metric=cfg["metric"])
    if a.dist_backend == "gloo":
        local = local % torch.cuda.device_count()                    # test mode: the ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if cpu_slice is not None and a.dist_backend == "nccl":          # one GPU per rank: move the slice next to this rank's GPU when sysfs says where that is
        try:
            ids = []
            for j in range(local_world):
                pr = torch.cuda.get_device_properties(j)
                ids.append("%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id))
            near = numa_slice(ids, local, set(allowed_cpus_at_start))
            if near:
                set_affinity_all_threads(near)                      # every thread that exists by now, not only this one
                cpu_slice = near
        except Exception:                                           # (placement is an optimisation: never a reason to fail the run)
            pass
    # torch.distributed: always for N > 1; for one rank when asked (--force-dist) or when a torchrun environment is present (WORLD_SIZE=1),
    # so that the RCCL group, the barriers and the two all-reduces of sharding.reduce_stats also run on a one-GPU box
    use_dist = world > 1 or a.force_dist or "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(free_port())
        form_group(a, cfg, rank, world, dev, ndev)

    import gfdm_amd
    from gfdm_amd import sharding, synth
    from gfdm_amd.filters import get_frequency_domain_filter

    K, M, L = cfg["K"], cfg["M"], cfg["L"]
    N = K * M
    use_eq, ic_iter, rx_bps, _ = MODES[cfg["mode"]]
    taps = get_frequency_domain_filter("rrc", cfg["alpha"], M, K, L)
    qpsk = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2)
    # the batched-blocks multi-GPU mode of the product (gfdm_amd.sharding.ShardedBatch): one kernel handle per device of this process
    # (one device per process under torchrun), contiguous shards, no data-path collective
    sb_mod = sharding.ShardedBatch(lambda d: gfdm_amd.Modulator(M, K, L, taps, device=d), [local], rank, world)
    sb_dem = sharding.ShardedBatch(lambda d: gfdm_amd.Demodulator(M, K, L, np.conj(taps), device=d), [local], rank, world)   # rx taps = conj(tx taps)
    sb_adv = sharding.ShardedBatch(lambda d: gfdm_amd.AdvancedReceiver(M, K, L, np.conj(taps), np.arange(K), 2, qpsk, device=d), [local], rank, world)
    mod, dem, adv = sb_mod.kernels[0], sb_dem.kernels[0], sb_adv.kernels[0]
    plan = shard_plan(cfg, a.batch, rank, world, sb_dem)
    B, _, total_per_step, scaling = plan
    L_ = gfdm_amd.lib()
    stream = torch.cuda.current_stream().cuda_stream
    buf_bytes = B * N * 8
    zeros3 = lambda: torch.zeros(3, dtype=torch.float64, device=dev)

    def slots(nbuf, batch_bytes=buf_bytes):
        return max(2, min(256, (a.ring_mib << 20) // (nbuf * batch_bytes)))

    def gen_symbols(block_start, count):
        return torch.cat([synth.qpsk_symbols(block_start + c, min(GEN_CHUNK, count - c), N, dev) for c in range(0, count, GEN_CHUNK)]) \
            if count > GEN_CHUNK else synth.qpsk_symbols(block_start, count, N, dev)

    def gen_rx_inputs(block_start, count, with_eq):
        """modulated frames (through the per-block test channel when with_eq) of `count` blocks from global block `block_start`"""
        fr = torch.empty(count, N, dtype=torch.complex64, device=dev)
        eq = torch.empty(count, N, dtype=torch.complex64, device=dev) if with_eq else None
        for c in range(0, count, GEN_CHUNK):
            n = min(GEN_CHUNK, count - c)
            x = mod.modulate(synth.qpsk_symbols(block_start + c, n, N, dev))
            if with_eq:
                eq[c:c + n] = synth.channel_response(block_start + c, n, N, dev)
                x = synth.through_test_channel(x, block_start + c)      # (element-wise: the same block whatever chunk or shard generates it)
            fr[c:c + n] = x
        torch.cuda.synchronize()
        return fr, eq

    rx_fn, sb_rx = ((L_.gfdm_hip_advanced_receiver_work_device, sb_adv) if ic_iter else (L_.gfdm_hip_receiver_demodulate_device, sb_dem))
    # kernel name as rocprofv3 prints it; last argument = IcKind
    ick = 0 if not ic_iter else (2 if (K >= 128 and 4 <= M <= 16) else 1)     # 1: real even IC kernel on the vector ALU, 2: on the matrix cores (K >= 128)
    rx_template = "k_row_receive<%d, %d, %d, %d, %d, %d>" % (K, M, L, 2 if ic_iter else 1, 1 if use_eq else 0, ick)
    mod_template = "k_row_modulate<%d, %d, %d, 0>" % (K, M, L)

    # ---- headline ----------------------------------------------------------------------------------------------------
    # Steps are independent batches (own ring slot each), so they are pipelined over `--streams` HIP streams: slot s always
    # runs on stream s % S, i.e. within a stream the kernels of a slot stay ordered while the load / compute / store phases of
    # neighbouring steps overlap on the GPU.  `value` is timed on that region.  The `roofline` objects are taken from a
    # single-stream replay of the same steps (one kernel on the GPU at a time).
    S = max(1, a.streams)
    two_kernel = cfg["mode"] == "mod_demod_mf"
    nbuf = 3 if (two_kernel or use_eq) else 2
    ns = max(S, (slots(nbuf) // S) * S) if scaling == "weak" else max(2, min(slots(nbuf), 4))
    if two_kernel:
        sym = [gen_symbols(slot_block_start(plan, rank, s), B) for s in range(ns)]
        frames = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(ns)]
        eqs = [None] * ns
    else:
        sym = None
        frames, eqs = zip(*[gen_rx_inputs(slot_block_start(plan, rank, s), B, use_eq) for s in range(ns)])
    outs = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(ns)]
    side = [torch.cuda.Stream(device=dev) for _ in range(S)]

    # cfg2's step is modulate + demodulate.  Demodulating the frames a step has just modulated reads 18.9 MB that may still sit in the
    # 256 MiB Infinity Cache; the headline therefore demodulates the frames modulated `lag` steps EARLIER (>= 256 MiB of other traffic in
    # between, the same stream, so the order holds): every step is still one modulate + one demodulate of a whole batch.  The same-slot
    # ("hot") figure is reported beside it.
    step_bytes = 32 * N * B if two_kernel else rx_bps * N * B
    lag = 0
    if two_kernel:
        lag = S * max(1, -(-(256 << 20) // (step_bytes * S)))
        if lag >= ns:
            lag = 0                                        # ring too small to separate the two kernels (tiny --ring-mib): hot only

    def step_fns_on(stream_of_slot, demod_lag=0):
        fns = []
        for s in range(ns):
            step = []
            if two_kernel:
                step.append(sb_mod.prepare(L_.gfdm_hip_modulator_work_device, [frames[s]], [(sym[s],)], [B], [stream_of_slot(s)]))
            r = (s - demod_lag) % ns
            step.append(sb_rx.prepare(rx_fn, [outs[r]], [(frames[r], eqs[r])], [B], [stream_of_slot(s)]))
            fns.append(step)
        return fns

    if two_kernel:                                         # every ring slot holds modulated frames before anything is timed
        for s in range(ns):
            sb_mod.prepare(L_.gfdm_hip_modulator_work_device, [frames[s]], [(sym[s],)], [B], [stream])()
        torch.cuda.synchronize()
    piped = step_fns_on(lambda s: side[s % S].cuda_stream, lag)
    launch_cpu = pin_launch_thread(cpu_slice)              # N > 1: the timed regions' launch loops run on one CPU of this rank's slice (timed_loop)
    wall = timed_loop(piped, a.steps, a.warmup, world, time_kernels=False)[0]
    total_blocks, _, wall_max = sharding.reduce_stats(B * a.steps, zeros3(), wall, dev)
    rank_wall_ms = gather_ranks(wall * 1e3, world)         # every rank's own time for its K steps: a shortfall at N > 1 can be read off one run
    value = total_blocks / wall_max
    value_hot = None
    if lag:
        hot = step_fns_on(lambda s: side[s % S].cuda_stream, 0)
        wall_h = timed_loop(hot, a.steps, a.warmup, world, time_kernels=False)[0]
        hot_blocks, _, wall_h_max = sharding.reduce_stats(B * a.steps, zeros3(), wall_h, dev)
        value_hot = hot_blocks / wall_h_max
        del hot
    sustained = None
    if a.sustained_seconds > 0:
        nsus = max(a.steps, int(a.sustained_seconds / max(wall_max / a.steps, 1e-7)) + 1)
        wsus = timed_loop(piped, nsus, 0, world, time_kernels=False)[0]
        sus_blocks, _, wsus_max = sharding.reduce_stats(B * nsus, zeros3(), wsus, dev)
        sustained = {"seconds": wsus_max, "steps": nsus, "value": sus_blocks / wsus_max, "ms_per_step": wsus_max / nsus * 1e3}
    single = step_fns_on(lambda s: stream, lag)
    wall1, kern_ms, kern_single, kern_sus = timed_loop(single, a.steps, a.warmup, world, time_kernels=True, kernel_seconds=a.kernel_seconds)
    _, _, wall1_max = sharding.reduce_stats(0, zeros3(), wall1, dev)
    # output checksum of ring slot 0 (strong scaling: the union over the ranks is global blocks [0, total) whatever N is)
    for f in step_fns_on(lambda s: stream, 0)[0]:
        f()
    torch.cuda.synchronize()
    _, chk, _ = sharding.reduce_stats(0, sharding.output_checksum(outs[0]), 0.0, dev)

    names = (["modulate"] if two_kernel else []) + [cfg["mode"] if not two_kernel else "demodulate"]
    templates = ([mod_template] if two_kernel else []) + [rx_template]
    bps = ([16] if two_kernel else []) + [rx_bps]
    rk = {}
    build_id = gfdm_amd.build_id()
    frac_of = lambda nbytes, ms: None if not ms else nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
    for nm, tp, bp, ms_piped, ms, ms_sus in zip(names, templates, bps, kern_ms, kern_single, kern_sus):
        ach = bp * N * B / (ms * 1e-3) / 1e9
        tr = pmc_traffic(tp, B, build_id)
        rp = rocprof_kernel_ms(tp, B, build_id)
        rk[nm] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                  "traffic": tr["bytes"], "traffic_source": tr.get("source"), "traffic_note": tr.get("note"),
                  "kernel": tp, "bytes_per_launch": bp * N * B, "kernel_ms": ms, "kernel_ms_pipelined": ms_piped,
                  "achieved_pipelined": bp * N * B / (ms_piped * 1e-3) / 1e9, "frac_pipelined": bp * N * B / (ms_piped * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                  "kernel_ms_sustained": ms_sus, "frac_sustained": frac_of(bp * N * B, ms_sus),
                  "kernel_ms_rocprofv3": rp["ms"], "frac_rocprofv3": frac_of(bp * N * B, rp["ms"]), "rocprofv3_source": rp.get("source"), "rocprofv3_note": rp.get("note")}
    dominant = max(rk, key=lambda k: rk[k]["kernel_ms"])
    roofline = dict(rk[dominant])
    roofline["kernel"] = "%s (%s, %s family)" % (roofline["kernel"], dominant, dem.kernel_name())
    roofline["copy_ceiling_GBps"] = copy_ceiling(roofline["bytes_per_launch"], a.steps, a.ring_mib, dev)
    roofline["region"] = ("kernel_ms / achieved / frac: median of HIP event pairs, one pair per single launch of this kernel over the ring slots (the pair itself "
                          "costs 1-2 us); *_pipelined: one event pair around the %d timed steps' launches back to back on one stream (a burst after idle); "
                          "*_sustained: the same back to back for %.1f s after a warm-up run = steady state; *_rocprofv3: mean kernel duration of the committed "
                          "rocprofv3 --kernel-trace collection of THIS build (null + rocprofv3_note when the committed collection is of another build)" % (a.steps, a.kernel_seconds))
    roofline["build_id"] = build_id
    if world > 1:
        roofline["rank"] = 0
    result = {
        "metric": cfg["metric"],
        "value": value, "unit": "blocks/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall_max / a.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        # cfg4's cancellation rounds run as v_mfma_f32_16x16x32_f16 on exact f16 decisions and a three-term f16 split of the IC taps (33
        # significant bits, not narrower than the reference's f32), everything else and every accumulation in f32
        "dtype": "f32 (IC rounds: f16x3-split MFMA, f32 accumulate)" if ick == 2 else "f32", "data": "synthetic",
        "config": {"workload": "%s; %d QPSK blocks per step %s, ring of %d buffer sets, independent steps pipelined over %d HIP streams%s"
                               % (cfg["workload"], total_per_step if scaling == "strong" else B,
                                  "in total, sharded contiguously over the GPUs" if scaling == "strong" else "per GPU", ns, S,
                                  "; roofline: rank 0's kernels, cpu_baseline and per-path figures only in the 1-GPU run" if world > 1 else ""),
                   "name": a.config, "block_size": N, "batch_per_gpu": B, "blocks_per_step_all_gpus": total_per_step, "streams": S,
                   "sharding": "independent blocks per GPU (gfdm_amd.sharding.ShardedBatch: one handle + stream per device, contiguous shards), no data-path collective"},
        "msym_per_s": value * N / 1e6,
        "value_single_stream": total_per_step * a.steps / wall1_max,
        # cfg2: `value` demodulates frames modulated `demod_lag_steps` steps earlier (cold: >= 256 MiB of traffic in between);
        # value_same_slot demodulates the frames its own step has just written (may be served by the Infinity Cache)
        "demod_lag_steps": lag, "value_same_slot": value_hot,
        "sustained": sustained,
        "roofline": roofline,
        "roofline_kernels": rk,
        "kernels": {"modulate": mod.kernel_name(), "demodulate": dem.kernel_name(), "advanced": adv.kernel_name()},
        "output_checksum": [float(v) for v in chk],
    }
    if use_dist:
        result["rank_wall_ms"] = rank_wall_ms
        result["rank_hosts_cpus"] = gather_ranks(rank_placement(cpu_slice, launch_cpu), world)
    del sym, frames, eqs, outs, piped, single

    # ---- per-variant measurements (each alone on the stream, own ring): single GPU, K=64 M=9 only ---------------------------
    want_paths = (not a.no_paths) and world == 1 and a.config in ("cfg2", "cfg3")
    if want_paths:
        paths = {}

        def measure(name, nbuf, bytes_per_sym, make_fns):
            n_slots = slots(nbuf)
            fns, keep = make_fns(n_slots)
            w, kms, ksingle, ksus = timed_loop(fns, a.steps, a.warmup, world, kernel_seconds=a.kernel_seconds)
            gbps = bytes_per_sym * N * B / (ksingle[0] * 1e-3) / 1e9
            paths[name] = {"blocks_per_s": B * a.steps / w, "msym_per_s": B * a.steps / w * N / 1e6,
                           "kernel_ms": ksingle[0], "kernel_ms_pipelined": kms[0], "kernel_ms_sustained": ksus[0],
                           "bytes_per_launch": int(round(bytes_per_sym * N * B)), "achieved_GBps": gbps,
                           "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS, "frac_of_hbm_peak_pipelined": bytes_per_sym * N * B / (kms[0] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                           "frac_of_hbm_peak_sustained": frac_of(bytes_per_sym * N * B, ksus[0]),
                           "ring_slots": n_slots}
            del keep

        gblock = lambda s: slot_block_start(plan, rank, s)

        def mk_mod(n_slots):
            i = [synth.qpsk_symbols(gblock(s), B, N, dev) for s in range(n_slots)]
            o = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(n_slots)]
            return [[raw_launcher(L_.gfdm_hip_modulator_work_device, mod._h, o[s], [i[s]], B, stream)] for s in range(n_slots)], (i, o)

        def mk_rx(fn, handle, with_eq):
            def make(n_slots):
                fr, eq = zip(*[gen_rx_inputs(gblock(s), B, with_eq) for s in range(n_slots)])
                o = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(n_slots)]
                return [[raw_launcher(fn, handle, o[s], [fr[s], eq[s]], B, stream)] for s in range(n_slots)], (fr, eq, o)
            return make

        measure("modulate", 2, 16, mk_mod)
        measure("demod_mf", 2, 16, mk_rx(L_.gfdm_hip_receiver_demodulate_device, dem._h, False))
        measure("demod_zf", 3, 24, mk_rx(L_.gfdm_hip_receiver_demodulate_device, dem._h, True))
        measure("demod_mf_ic2", 2, 16, mk_rx(L_.gfdm_hip_advanced_receiver_work_device, adv._h, False))
        measure("demod_zf_ic2", 3, 24, mk_rx(L_.gfdm_hip_advanced_receiver_work_device, adv._h, True))

        # the receive chain of examples/hier_gfdm_receiver.grc in ONE kernel: channel estimate from each frame's received
        # preamble + ZF + 2 IC + resource demapper (52 of 64 subcarriers active): 8 N + 16 K bytes read, 8 A M written per frame
        A_ = (52 * K) // 64
        smap_ = np.concatenate((np.arange(1, A_ // 2 + 1), np.arange(K - A_ // 2, K)))
        pre_ = np.tile(np.fft.ifft(np.exp(2j * np.pi * np.random.default_rng(0).random(K))) * np.sqrt(K), 2)
        est_ = gfdm_amd.ChannelEstimator(M, K, A_, True, 1, pre_, device=local)
        advf = gfdm_amd.AdvancedReceiver(M, K, L, np.conj(taps), smap_, 2, qpsk, device=local)
        advf.configure_frames(N, 0, smap_, True)
        advf.set_channel_estimator(est_)

        def mk_chain(n_slots):
            fr = [gen_rx_inputs(gblock(s), B, False)[0] for s in range(n_slots)]
            rp = [torch.tensor(np.tile(pre_, (B, 1)), dtype=torch.complex64, device=dev) for _ in range(n_slots)]
            o = [torch.empty(B, A_ * M, dtype=torch.complex64, device=dev) for _ in range(n_slots)]

            def launcher(s):
                args = [advf._h, ctypes.c_void_p(o[s].data_ptr()), ctypes.c_void_p(fr[s].data_ptr()), ctypes.c_void_p(rp[s].data_ptr()),
                        ctypes.c_int(0), ctypes.c_int(-1), ctypes.c_int64(B), ctypes.c_void_p(stream)]

                def go():
                    rc = L_.gfdm_hip_advanced_receiver_work_estimated_device(*args)
                    if rc != 0:
                        raise RuntimeError("gfdm_hip launch failed: %d" % rc)
                return go
            return [[launcher(s)] for s in range(n_slots)], (fr, rp, o)

        measure("frames_zf_ic2_estimated", 2, (8.0 * N + 16.0 * K + 8.0 * A_ * M) / N, mk_chain)
        if not a.no_host_paths:
            paths.update(host_batch_paths(cfg, taps, [int(x) for x in a.host_sizes.split(",")], with_cpu=not a.no_cpu_baseline))
        result["paths"] = paths
        ns_tp = "k_row_receive<%d, %d, %d, 2, 1, 1>" % (K, M, L)
        ns_rp = rocprof_kernel_ms(ns_tp, B, build_id)
        ns = paths["demod_zf_ic2"]
        result["north_star"] = {"path": "demod_zf_ic2 (BASELINE configs[2]: ZF demod + 2 IC iterations)",
                                "frac_of_hbm_peak": ns["frac_of_hbm_peak"], "target": 0.40}
        # inside `roofline` as well (the driver's record keeps that object whole): the north-star kernel on this box, all four readings
        result["roofline"]["north_star"] = {"kernel": ns_tp + " (BASELINE configs[2]: ZF demod + 2 IC iterations, %d blocks per launch)" % B, "target_frac": 0.40,
                                            "bytes_per_launch": ns["bytes_per_launch"], "kernel_ms": ns["kernel_ms"], "frac": ns["frac_of_hbm_peak"],
                                            "kernel_ms_pipelined": ns["kernel_ms_pipelined"], "frac_pipelined": ns["frac_of_hbm_peak_pipelined"],
                                            "kernel_ms_sustained": ns["kernel_ms_sustained"], "frac_sustained": ns["frac_of_hbm_peak_sustained"],
                                            "kernel_ms_rocprofv3": ns_rp["ms"], "frac_rocprofv3": frac_of(ns["bytes_per_launch"], ns_rp["ms"]),
                                            "rocprofv3_source": ns_rp.get("source"), "rocprofv3_note": ns_rp.get("note")}

    # ---- the same kernels at the batch size of BASELINE configs[3,4] (65 536 blocks per launch): steady-state roofline --------
    if a.large_batch > 0 and want_paths:
        BL = a.large_batch
        large = {}
        nsl = 3
        for name, fn, handle, with_eq, bps_ in (("demod_mf", L_.gfdm_hip_receiver_demodulate_device, dem._h, False, 16),
                                                ("demod_zf_ic2", L_.gfdm_hip_advanced_receiver_work_device, adv._h, True, 24)):
            fr, eq = zip(*[gen_rx_inputs((1000 + sl) * BL, BL, with_eq) for sl in range(nsl)])
            o = [torch.empty(BL, N, dtype=torch.complex64, device=dev) for _ in range(nsl)]
            fns = [[raw_launcher(fn, handle, o[sl], [fr[sl], eq[sl]], BL, stream)] for sl in range(nsl)]
            # three readings of the same kernel in one run: (1) a burst of 10 launches right after the set-up, one event pair around it -- what
            # round 4 reported as kernel_ms; it depends on the clock state the idle gap left behind (0.54 of peak on one box, 0.73 on the next);
            # (2) back to back for >= 1 s after a warm-up run: the steady state; (3) the median of per-launch event pairs over an equally long run.
            # (2) and (3) agree within 2 % (profiles/r05/sustained_vs_burst.txt); rocprofv3's kernel duration -- one kernel on the GPU at a time, no overlap of one launch's
            # tail with the next one's head -- reads up to 10 % longer.
            torch.cuda.synchronize()
            eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            eb0.record()
            for i in range(10):
                fns[i % nsl][0]()
            eb1.record()
            torch.cuda.synchronize()
            burst_ms = eb0.elapsed_time(eb1) / 10
            sus_ms, nsus_l = sustained_launch_ms(lambda i: fns[i % nsl][0](), 1.0, burst_ms)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nsus_l)]
            for i in range(nsus_l):
                evs[i][0].record()
                fns[i % nsl][0]()
                evs[i][1].record()
            torch.cuda.synchronize()
            per_launch = sorted(x.elapsed_time(y) for x, y in evs)
            kms_med = float(np.median(per_launch))
            tp = "k_row_receive<%d, %d, %d, %d, %d, %d>" % (K, M, L, 2 if with_eq else 1, 1 if with_eq else 0, 1 if with_eq else 0)
            rp = rocprof_kernel_ms(tp, BL, build_id)
            nbytes = bps_ * N * BL
            large[name] = {"blocks_per_launch": BL, "kernel": tp, "bytes_per_launch": nbytes, "blocks_per_s": BL / (sus_ms * 1e-3),
                           "kernel_ms": sus_ms, "frac_of_hbm_peak": frac_of(nbytes, sus_ms), "achieved_GBps": nbytes / (sus_ms * 1e-3) / 1e9,
                           "launches_sustained": nsus_l,
                           "kernel_ms_per_launch_median": kms_med, "frac_of_hbm_peak_per_launch_median": frac_of(nbytes, kms_med),
                           "kernel_ms_per_launch_p10_p90": [per_launch[len(per_launch) // 10], per_launch[(9 * len(per_launch)) // 10]],
                           "kernel_ms_burst_after_idle": burst_ms, "frac_of_hbm_peak_burst_after_idle": frac_of(nbytes, burst_ms),
                           "kernel_ms_rocprofv3": rp["ms"], "frac_of_hbm_peak_rocprofv3": frac_of(nbytes, rp["ms"]), "rocprofv3_note": rp.get("note")}
            del fr, eq, o, fns
        result["large_batch"] = large

    # the driver's record keeps the SCALAR keys of `roofline` (nested objects are dropped, other top-level keys survive by name only): the north-star kernel's
    # readings and the sustained headline therefore go in flat, beside the nested objects kept for human readers
    if sustained:
        result["sustained_value"] = sustained["value"]
        result["roofline"]["value_sustained"] = sustained["value"]
        result["roofline"]["value_sustained_seconds"] = sustained["seconds"]
    if want_paths:
        nsd = result["roofline"]["north_star"]
        flat = {"north_star_kernel": nsd["kernel"], "north_star_target_frac": 0.40, "north_star_kernel_ms": nsd["kernel_ms"], "north_star_frac": nsd["frac"],
                "north_star_frac_pipelined": nsd["frac_pipelined"], "north_star_frac_sustained": nsd["frac_sustained"],
                "north_star_kernel_ms_rocprofv3": nsd["kernel_ms_rocprofv3"], "north_star_frac_rocprofv3": nsd["frac_rocprofv3"]}
        lb = result.get("large_batch", {}).get("demod_zf_ic2")
        if lb:
            n_ = lb["blocks_per_launch"]                   # 65 536 unless --large-batch says otherwise
            flat.update({"north_star_frac_%d_sustained" % n_: lb["frac_of_hbm_peak"], "north_star_frac_%d_per_launch_median" % n_: lb["frac_of_hbm_peak_per_launch_median"],
                         "north_star_frac_%d_rocprofv3" % n_: lb["frac_of_hbm_peak_rocprofv3"], "north_star_blocks_per_s_%d_sustained" % n_: lb["blocks_per_s"]})
        result["roofline"].update(flat)

    if rank == 0 and world == 1:
        result["single_block_host_us"] = single_block_host(cfg, np.conj(taps))
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cfg, taps, a.cpu_seconds)
    elif rank == 0:
        result["cpu_baseline"] = None
    pr = torch.cuda.get_device_properties(local)
    rank_devices = gather_ranks({"rank": rank, "device": local, "name": pr.name,
                                 "pci": "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))}, world)
    if rank == 0:
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                                                 # noqa: BLE001
            rccl = None
        result["distributed"] = {"initialized": bool(use_dist), "backend": (a.dist_backend + (" (RCCL)" if a.dist_backend == "nccl" else "")) if use_dist else None,
                                 "world_size": world, "rccl_version": rccl, "devices_visible": ndev, "rank_devices": rank_devices,
                                 "collectives": "2 barriers per timed loop + 2 all-reduces per statistic (sharding.reduce_stats); no payload collective",
                                 "scaling_curve": "none measured so far: no multi-GPU node was available to the builder or the driver in rounds 1-6" if world == 1 else "this line is one point of it"}
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
