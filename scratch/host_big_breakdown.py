"""Time breakdown (gfdm_hip_host_call_times) of large pageable host calls against copy threads / chunk size: is the pipeline host- or link-bound?"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import gfdm_amd as g
from gfdm_amd.filters import get_frequency_domain_filter
M, K, L = 9, 64, 2; N = M * K
dem = g.Demodulator(M, K, L, get_frequency_domain_filter("rrc", 0.2, M, K, L))
nb = 32768
x = (np.random.default_rng(0).standard_normal((nb, N)) + 0j).astype(np.complex64)
out = np.empty_like(x)
for stream_copies, thr in [(sc, t) for t in (0, 1, 2, 3, 5) for sc in (0, 1)]:
    g.lib().gfdm_hip_set_host_streaming_copies_for_testing(stream_copies)
    for chunk in (4 << 20, 16 << 20):
        g.set_host_pipeline(0, chunk, 3, thr, 2)
        for _ in range(2): dem.demodulate(x, out=out)
        t0 = time.perf_counter(); dem.demodulate(x, out=out); dt = time.perf_counter() - t0
        st = g.host_call_stats()
        print(("streaming stores, " if stream_copies else "plain memcpy,     ") + "threads %d chunk %2d MiB: %.2f M blocks/s, call %.0f us = copy %.0f + launch %.0f + post %.0f + wait %.0f us; chunks %d, pool threads used %d; copy rate %.1f GB/s" %
              (thr, chunk >> 20, nb / dt / 1e6, dt * 1e6, st["ns"]["copy"] / 1e3, st["ns"]["launch"] / 1e3, st["ns"]["post"] / 1e3, st["ns"]["wait"] / 1e3, st["chunks"], st["copy_threads"],
               2 * x.nbytes / (st["ns"]["copy"] * 1e-9) / 1e9))
