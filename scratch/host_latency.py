"""Latency of the host-pointer path (generic_work through the pybind11 drop-in class) against the batch size."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "lib"))
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
import gfdm_amd          # ctypes path: honours GFDM_HIP_LIB (A/B builds); the pybind11 module always loads the in-tree library
from gfdm_amd.filters import get_frequency_domain_filter
M, K, L = 9, 64, 2
taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
dem = gfdm_amd.Demodulator(M, K, L, taps)
adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2))
rng = np.random.default_rng(0)
for nb in (1, 4, 16, 32, 64, 128, 256, 512, 1024, 4096):
    x = (rng.standard_normal(nb * M * K) + 1j * rng.standard_normal(nb * M * K)).astype(np.complex64)
    f = np.ones_like(x)
    res = []
    for name, fn in (("demodulate", lambda: dem.demodulate(x)), ("zf+ic2", lambda: adv.demodulate_equalize(x, f))):
        for _ in range(10): fn()
        reps = 200 if nb <= 64 else 30
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        res.append("%s %8.1f us (%6.2f us/block)" % (name, (time.perf_counter() - t0) / reps * 1e6, (time.perf_counter() - t0) / reps * 1e6 / nb))
    print("blocks %5d  " % nb + "   ".join(res))
