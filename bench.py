#!/usr/bin/env python3
"""bench.py -- GFDM blocks/s on MI355X for BASELINE.json configs[1]: K=64 subcarriers, M=9 timeslots, RRC alpha=0.2,
overlap=2, batch = 4096 blocks per step, modulate + matched-filter demodulate ("mod+demod").

One STEP = one pass of the hot path over one batch that is already resident in HBM:
    frames = modulate(symbols)            (HIP kernel 1)
    out    = demodulate(frames)           (HIP kernel 2, the dominant one: `roofline` describes it)
Steps rotate through a ring of distinct device buffers (default >= 2 GiB) so that consecutive steps cannot be
served from the 256 MiB Infinity Cache.  W warm-up steps, then exactly K timed steps bracketed by barrier +
synchronize on both sides; the time is the MAX over ranks; `value` = all blocks all ranks processed / that time.
Every rank processes its own 4096-block batches (weak scaling, no data-path collective: GFDM blocks are independent).

Beside the headline the JSON line carries
  roofline      achieved HBM GB/s of the dominant kernel = algorithmic bytes per launch (16 B/symbol for MF,
                24 B/symbol with the per-block equaliser vector) / its mean duration measured with
                HIP events on the launch stream around every launch of the timed region (this includes ~2-3 us of
                dispatch latency per launch that rocprofv3's kernel trace does not count: profiles/README.md)
  cpu_baseline  the plain-C oracle ("port" of the reference algorithm, oracle/gfdm_oracle.c) timed on this host's cores
                on a bounded sample of the same workload (rank 0, N=1 only)
  paths         the same measurement for each receiver variant (MF, ZF, ZF + 2 IC iterations = configs[2], the
                north-star path) and the modulator, each over its own ring
"""
import argparse
import ctypes
import ctypes.util
import gc
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy reaches

CFG = dict(K=64, M=9, L=2, alpha=0.2)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="GFDM blocks per step and per GPU (configs[1]: 4096)")
    ap.add_argument("--ring-mib", type=int, default=2048, help="total footprint of the buffer ring per path")
    ap.add_argument("--streams", type=int, default=4, help="HIP streams the independent steps of the headline loop are pipelined over")
    ap.add_argument("--large-batch", type=int, default=65536, help="blocks per launch of the extra large-batch measurement (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-paths", action="store_true", help="skip the per-variant measurements")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU time of each cpu_baseline leg")
    return ap.parse_args()


class Ring:
    """Ring of device buffers for one path; slot s holds a full batch of inputs and its output."""

    def __init__(self, nslots, make_inputs, batch, N, device):
        self.inputs = [make_inputs(s) for s in range(nslots)]
        self.outs = [torch.empty(batch, N, dtype=torch.complex64, device=device) for _ in range(nslots)]
        self.n = nslots


def raw_launcher(fn, handle, out_t, in_ts, nblocks, stream_ptr):
    """Pre-marshalled ctypes call (keeps per-launch host cost at ~1-2 us)."""
    args = [handle, ctypes.c_void_p(out_t.data_ptr())] + [ctypes.c_void_p(t.data_ptr()) if t is not None else None for t in in_ts]
    args += [ctypes.c_int64(nblocks), ctypes.c_void_p(stream_ptr)]

    def go():
        rc = fn(*args)
        if rc != 0:
            raise RuntimeError("gfdm_hip launch failed: %d" % rc)
    return go


def timed_loop(step_fns, dominant, steps, warmup, world, time_kernel=True):
    """Run warmup + `steps` timed steps.  step_fns[i] is the list of launch closures of ring slot i (each closure is bound
    to its stream); `dominant` is the index (within a step) of the kernel the roofline describes.  With `time_kernel`
    (single-stream loops only) ONE HIP event pair on the launch stream brackets a back-to-back run of the `steps` launches
    of that kernel, i.e. the launch-to-launch duration: the kernel itself plus the ~1.3 us dispatch gap between two
    dependent launches (an event pair around every single launch would add another ~2.6 us of event packets to each).
    Returns (wall seconds of the timed region, mean duration in ms of the dominant kernel or None)."""
    nslots = len(step_fns)
    for i in range(warmup):
        for f in step_fns[i % nslots]:
            f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gc.collect()
    gc.disable()                              # no collector pause (tens of ms with the tensor rings alive) inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if time_kernel:
        e0.record()
    for i in range(steps):
        for f in step_fns[(warmup + i) % nslots]:
            f()
    if time_kernel:
        e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0           # this rank's time for its K steps; the caller takes the MAX over ranks
    gc.enable()
    if world > 1:
        dist.barrier()                        # closing bracket: every rank has finished before anyone moves on
    torch.cuda.synchronize()
    kern_ms = None
    if time_kernel:
        if len(step_fns[0]) > 1:          # several kernels per step: replay only the dominant one over the same ring slots
            e0.record()
            for i in range(steps):
                step_fns[(warmup + i) % nslots][dominant]()
            e1.record()
            torch.cuda.synchronize()
        kern_ms = e0.elapsed_time(e1) / steps
    return wall, kern_ms


def cpu_baseline(taps, batch_blocks, seconds):
    """Time the plain-C oracle (reference algorithm, restated; FFTW/VOLK are not installed) on mod + MF demod.
    Leg 1: one kernel object, one thread, block by block (how simple_receiver_cc_impl::work drives the reference).
    Leg 2: T threads, one kernel object each, disjoint block ranges, T = the cores this process may run on.
    Both legs are time-bounded (`seconds` each) so the default bench run stays within minutes."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    try:    # rebuild for this host's ISA; fall back to the shipped portable build
        path = "/tmp/libgfdm_oracle_native_%d.so" % os.getpid()
        c_oracle.build(cflags=["-O3", "-march=native"], out=path)
        lib = c_oracle.load(path)
    except Exception:
        lib = c_oracle.load()
    K, M, L = CFG["K"], CFG["M"], CFG["L"]
    N = K * M
    rng = np.random.default_rng(0)
    nb = 1024
    sym = (((1 - 2 * rng.integers(0, 2, (nb, N))) + 1j * (1 - 2 * rng.integers(0, 2, (nb, N)))) / np.sqrt(2)).astype(np.complex64)

    def run(o, deadline, counter, idx):
        n = 0
        while True:
            o.demodulate(o.modulate(sym))
            n += nb
            if time.perf_counter() >= deadline:
                break
        counter[idx] = n

    def leg(nthreads):
        objs = [c_oracle.COracle(M, K, L, taps, lib=lib) for _ in range(nthreads)]
        counter = [0] * nthreads
        t0 = time.perf_counter()
        deadline = t0 + seconds
        threads = [threading.Thread(target=run, args=(objs[i], deadline, counter, i)) for i in range(nthreads)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        return sum(counter) / (time.perf_counter() - t0), sum(counter)

    single, n1 = leg(1)
    try:
        T = len(os.sched_getaffinity(0))
    except AttributeError:
        T = os.cpu_count() or 1
    T = max(1, min(T, 64))
    multi, nT = leg(T)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": multi, "unit": "blocks/s", "cores": T, "kind": "port",
            "sample": "mod + MF demod of %d QPSK blocks (K=64 M=9 L=2) in %.0f s on %d threads, one plain-C oracle kernel object "
                      "per thread (-O3 -march=native); single thread: %d blocks in %.0f s" % (nT, seconds, T, n1, seconds),
            "single_thread_value": single, "cpu_model": model, "host_logical_cpus": os.cpu_count(),
            # the reference's own CPU kernels need FFTW3f and VOLK; neither is installed on the boxes of this pool (probe),
            # so the only CPU figure is the plain-C restatement of the same per-block algorithm
            "reference_libs_present": {"fftw3f": ctypes.util.find_library("fftw3f") is not None, "volk": ctypes.util.find_library("volk") is not None}}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernels have no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL; only barriers/stat reductions

    import gfdm_amd
    from gfdm_amd import sharding, synth
    from gfdm_amd.filters import get_frequency_domain_filter

    K, M, L = CFG["K"], CFG["M"], CFG["L"]
    N, B = K * M, a.batch
    taps = get_frequency_domain_filter("rrc", CFG["alpha"], M, K, L)
    mod = gfdm_amd.Modulator(M, K, L, taps, device=local)
    dem = gfdm_amd.Demodulator(M, K, L, np.conj(taps), device=local)          # rx taps = conj(tx taps)  (matched filter)
    qpsk = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, np.conj(taps), np.arange(K), 2, qpsk, device=local)
    L_ = gfdm_amd.lib()
    stream = torch.cuda.current_stream().cuda_stream
    buf_bytes = B * N * 8

    def slots(nbuf):
        return max(2, min(256, (a.ring_mib << 20) // (nbuf * buf_bytes)))

    gblock = lambda s: (rank * 1000003 + s) * B          # distinct global block range per rank and slot

    # ---- headline: mod + MF demod -------------------------------------------------------------------------------
    # Steps are independent batches (own ring slot each), so they are pipelined over `--streams` HIP streams: slot s always
    # runs on stream s % S, i.e. within a stream modulate -> demodulate of a slot stay ordered while the load / compute /
    # store phases of neighbouring steps overlap on the GPU.  `value` is timed on that region.  The `roofline` object is
    # taken from a single-stream replay of the same steps (one kernel on the GPU at a time): a HIP event pair around the
    # back-to-back run of the timed steps' demodulate launches (timed_loop).
    S = max(1, a.streams)
    ns = max(S, (slots(3) // S) * S)
    sym = [synth.qpsk_symbols(gblock(s), B, N, dev) for s in range(ns)]
    frames = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(ns)]
    outs = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(ns)]
    side = [torch.cuda.Stream(device=dev) for _ in range(S)]

    def step_fns_on(stream_of_slot):
        return [[raw_launcher(L_.gfdm_hip_modulator_work_device, mod._h, frames[s], [sym[s]], B, stream_of_slot(s)),
                 raw_launcher(L_.gfdm_hip_receiver_demodulate_device, dem._h, outs[s], [frames[s], None], B, stream_of_slot(s))]
                for s in range(ns)]

    wall, _ = timed_loop(step_fns_on(lambda s: side[s % S].cuda_stream), 1, a.steps, a.warmup, world, time_kernel=False)
    chk = sharding.output_checksum(outs[(a.warmup + a.steps - 1) % ns])
    total_blocks, chk, wall_max = sharding.reduce_stats(B * a.steps, chk, wall, dev)
    value = total_blocks / wall_max
    wall1, kern_ms = timed_loop(step_fns_on(lambda s: stream), 1, a.steps, a.warmup, world, time_kernel=True)
    _, _, wall1_max = sharding.reduce_stats(0, torch.zeros(3, dtype=torch.float64, device=dev), wall1, dev)
    achieved = 16.0 * N * B / (kern_ms * 1e-3) / 1e9
    result = {
        "metric": "GFDM blocks/s, K=64 M=9 mod+demod (MF), batch 4096 per GPU",
        "value": value, "unit": "blocks/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall_max / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: K=64 subcarriers, M=9 timeslots, RRC alpha=0.2, overlap=2, MF receiver, "
                               "%d QPSK blocks per step per GPU, step = modulate + demodulate, ring of %d buffer sets, "
                               "independent steps pipelined over %d HIP streams" % (B, ns, S),
                   "block_size": N, "batch_per_gpu": B, "streams": S,
                   "sharding": "independent blocks per GPU, no data-path collective"},
        "msym_per_s": value * N / 1e6,
        "value_single_stream": world * B * a.steps / wall1_max,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x 2 [gfx950 correction, calibrated on a
                     # copy kernel of the same access shape] + WRITE_SIZE; profiles/r01/pmc_hbm_traffic_summary.csv).
                     # Counters cannot be read from inside this process, so the figure is only quoted for the
                     # configuration it was measured on.
                     "traffic": (18576 + 18432) * 1024 if (B == 4096 and dem.kernel_name() == "rowlane") else None,
                     "kernel": dem.kernel_name() + " (demodulate, MF)", "bytes_per_launch": 16 * N * B,
                     "kernel_ms": kern_ms,
                     "region": "single-stream replay of the %d timed steps' demodulate launches back to back, one HIP event pair around "
                               "the run (launch-to-launch time: kernel + dispatch gap; rocprofv3 kernel time: profiles/README.md)" % a.steps},
        "kernels": {"modulate": mod.kernel_name(), "demodulate": dem.kernel_name(), "advanced": adv.kernel_name()},
        "output_checksum": [float(v) for v in chk],
    }
    del sym, frames, outs

    # ---- per-variant measurements (each alone on the stream, own ring) ----------------------------------------------
    if not a.no_paths:
        paths = {}

        def measure(name, nbuf, bytes_per_sym, make_fns):
            n_slots = slots(nbuf)
            fns, keep = make_fns(n_slots)
            w, kms = timed_loop(fns, 0, a.steps, a.warmup, world)
            _, _, wmax = sharding.reduce_stats(0, torch.zeros(3, dtype=torch.float64, device=dev), w, dev)
            gbps = bytes_per_sym * N * B / (kms * 1e-3) / 1e9
            paths[name] = {"blocks_per_s": world * B * a.steps / wmax, "msym_per_s": world * B * a.steps / wmax * N / 1e6,
                           "kernel_ms": kms, "bytes_per_launch": int(round(bytes_per_sym * N * B)), "achieved_GBps": gbps,
                           "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS, "ring_slots": n_slots}
            del keep

        def rx_inputs(n_slots, with_eq):
            fr, eq = [], []
            for s in range(n_slots):
                x = mod.modulate(synth.qpsk_symbols(gblock(s), B, N, dev))
                if with_eq:
                    f = synth.channel_response(gblock(s), B, N, dev)
                    x = synth.through_channel(x, f)
                    eq.append(f)
                fr.append(x)
            torch.cuda.synchronize()
            return fr, eq

        def mk_mod(n_slots):
            i = [synth.qpsk_symbols(gblock(s), B, N, dev) for s in range(n_slots)]
            o = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(n_slots)]
            return [[raw_launcher(L_.gfdm_hip_modulator_work_device, mod._h, o[s], [i[s]], B, stream)] for s in range(n_slots)], (i, o)

        def mk_rx(fn, handle, with_eq):
            def make(n_slots):
                fr, eq = rx_inputs(n_slots, with_eq)
                o = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(n_slots)]
                return [[raw_launcher(fn, handle, o[s], [fr[s], eq[s] if with_eq else None], B, stream)] for s in range(n_slots)], (fr, eq, o)
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
            fr, _ = rx_inputs(n_slots, False)
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
        result["paths"] = paths
        result["north_star"] = {"path": "demod_zf_ic2 (BASELINE configs[2]: ZF demod + 2 IC iterations)",
                                "frac_of_hbm_peak": paths["demod_zf_ic2"]["frac_of_hbm_peak"], "target": 0.40}

    # ---- the same kernels at the batch size of BASELINE configs[3,4] (65 536 blocks per launch): steady-state roofline --------
    if a.large_batch > 0 and not a.no_paths:
        BL = a.large_batch
        large = {}
        nsl = 3
        for name, fn, handle, with_eq, bps in (("demod_mf", L_.gfdm_hip_receiver_demodulate_device, dem._h, False, 16),
                                               ("demod_zf_ic2", L_.gfdm_hip_advanced_receiver_work_device, adv._h, True, 24)):
            fr, eq, o = [], [], []
            for sl in range(nsl):
                x = mod.modulate(synth.qpsk_symbols(gblock(1000 + sl) , BL, N, dev))
                if with_eq:
                    f = synth.channel_response(gblock(1000 + sl), BL, N, dev)
                    x = synth.through_channel(x, f)
                    eq.append(f)
                fr.append(x)
                o.append(torch.empty(BL, N, dtype=torch.complex64, device=dev))
            torch.cuda.synchronize()
            fns = [[raw_launcher(fn, handle, o[sl], [fr[sl], eq[sl] if with_eq else None], BL, stream)] for sl in range(nsl)]
            nst = max(10, a.steps // 8)
            w, kms = timed_loop(fns, 0, nst, 3, world)
            _, _, wmax = sharding.reduce_stats(0, torch.zeros(3, dtype=torch.float64, device=dev), w, dev)
            # beside the back-to-back mean, the median of per-launch event pairs: a sustained run of launches of this size makes some
            # boxes of the pool drop their clocks after a few milliseconds (power cap), and idle gaps make them ramp down as well, so
            # the two figures bracket the kernel's duration
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nst)]
            for i in range(nst):
                evs[i][0].record()
                fns[i % nsl][0]()
                evs[i][1].record()
            torch.cuda.synchronize()
            kms_med = float(np.median([x.elapsed_time(y) for x, y in evs]))
            gbps = bps * N * BL / (kms * 1e-3) / 1e9
            large[name] = {"blocks_per_launch": BL, "blocks_per_s": world * BL * nst / wmax, "kernel_ms": kms,
                           "kernel_ms_per_launch_median": kms_med,
                           "bytes_per_launch": bps * N * BL, "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS}
            del fr, eq, o, fns
        result["large_batch"] = large

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(taps, B, a.cpu_seconds)
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
