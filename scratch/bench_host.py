#!/usr/bin/env python3
"""Throughput of the host-buffer batch path (*_host entry points) at K=64 M=9: blocks/s and bytes over the PCIe link per second for
1 ... 65 536 blocks per call, pageable numpy memory (bounced) and registered memory (in place), kernels across the link (mode 0) against
copy engines (mode 1), 0-3 copy threads, several chunk sizes.  Round-4 tuning tool; bench.py carries the chosen configuration.

    python3 scratch/bench_host.py [--sweep] [--sizes 1,16,256,4096,65536]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1,16,256,4096,65536")
    ap.add_argument("--sweep", action="store_true", help="chunk size / copy thread / mode sweep at 4096 and 65536 blocks")
    ap.add_argument("--seconds", type=float, default=0.4)
    a = ap.parse_args()
    import gfdm_amd
    import gfdm_ref as R
    from gfdm_amd.filters import get_frequency_domain_filter
    M, K, L = 9, 64, 2
    N = M * K
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    mod = gfdm_amd.Modulator(M, K, L, taps)
    dem = gfdm_amd.Demodulator(M, K, L, np.conj(taps))
    adv = gfdm_amd.AdvancedReceiver(M, K, L, np.conj(taps), np.arange(K), 2, R.qpsk_points())
    Lb = gfdm_amd.lib()
    sizes = [int(s) for s in a.sizes.split(",")]
    nmax = max(sizes)
    rng = np.random.default_rng(0)
    sym = gfdm_amd.aligned_copy(((1 - 2 * rng.integers(0, 2, (nmax, N))) + 1j * (1 - 2 * rng.integers(0, 2, (nmax, N)))).astype(np.complex64) / np.float32(np.sqrt(2)))
    frames = gfdm_amd.aligned_copy(mod.modulate(sym))
    feq = gfdm_amd.aligned_empty((nmax, N)); feq[...] = 1.0
    out = gfdm_amd.aligned_empty((nmax, N))
    paths = {
        "modulate": (Lb.gfdm_hip_modulator_work_host, mod._h, (sym,), 16),
        "demod_mf": (Lb.gfdm_hip_receiver_demodulate_host, dem._h, (frames, None), 16),
        "zf_ic2": (Lb.gfdm_hip_advanced_receiver_work_host, adv._h, (frames, feq), 24),
    }

    def rate(name, nb, seconds=a.seconds):
        fn, h, ins, bps = paths[name]
        args = [h, ctypes.c_void_p(out.ctypes.data)] + [None if x is None else ctypes.c_void_p(x.ctypes.data) for x in ins] + [ctypes.c_int64(nb)]
        if name == "modulate":
            args = [h, ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(ins[0].ctypes.data), ctypes.c_int64(nb)]
        for _ in range(2):
            assert fn(*args) == 0
        n, t0 = 0, time.perf_counter()
        while True:
            assert fn(*args) == 0
            n += 1
            dt = time.perf_counter() - t0
            if dt > seconds and n >= 3:
                break
        return {"blocks_per_s": nb * n / dt, "link_GBps": bps * N * nb * n / dt / 1e9, "us_per_call": dt / n * 1e6, "stats": gfdm_amd.host_call_stats()}

    res = {"build_id": gfdm_amd.build_id(), "default_pipeline": gfdm_amd.get_host_pipeline(), "pageable": {}, "registered": {}}
    for name in paths:
        res["pageable"][name] = {nb: rate(name, nb) for nb in sizes}
    with gfdm_amd.registered_host(sym, frames, feq, out):
        for name in paths:
            res["registered"][name] = {nb: rate(name, nb) for nb in sizes}
    if a.sweep:
        sw = {}
        for nb in (256, 4096, 65536):
            if nb > nmax:
                continue
            for mode in (0, 1, 2, 3):
                for threads in (2, 3, 5):
                    for chunk_kib in (1024, 4096, 16384):
                        for streams in ((1, 2) if mode == 0 else (1,)):
                            if nb == 256 and chunk_kib > 1024:
                                continue
                            gfdm_amd.set_host_pipeline(mode, chunk_kib << 10, 3, threads, streams)
                            for name in ("demod_mf", "zf_ic2"):
                                r = rate(name, nb, 0.25)
                                sw["pageable %s nb=%d mode=%d streams=%d threads=%d chunk=%dKiB" % (name, nb, mode, streams, threads, chunk_kib)] = round(r["blocks_per_s"] / 1e6, 3)
            with gfdm_amd.registered_host(sym, frames, feq, out):
                for mode in (0, 1, 2, 3):
                    for chunk_kib in (1024, 4096, 16384):
                        if mode == 0 and chunk_kib > 1024:
                            continue
                        gfdm_amd.set_host_pipeline(mode, chunk_kib << 10, 3, 2, 1)
                        for name in ("demod_mf", "zf_ic2"):
                            r = rate(name, nb, 0.25)
                            sw["registered %s nb=%d mode=%d chunk=%dKiB" % (name, nb, mode, chunk_kib)] = round(r["blocks_per_s"] / 1e6, 3)
        gfdm_amd.set_host_pipeline(0, 0, 3, 2, 2)
        res["sweep_Mblocks_per_s"] = sw
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
