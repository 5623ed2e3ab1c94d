"""Where a *_host call spends its time (gfdm_hip_host_call_times), K=64 M=9 MF demodulation, median over 2000 calls per size; pageable / registered."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import gfdm_amd
from gfdm_amd.filters import get_frequency_domain_filter
M, K, L = 9, 64, 2
N = M * K
dem = gfdm_amd.Demodulator(M, K, L, get_frequency_domain_filter("rrc", 0.2, M, K, L))
x = gfdm_amd.aligned_copy((np.random.default_rng(0).standard_normal((256, N)) + 0j).astype(np.complex64))
out = gfdm_amd.aligned_empty(x.shape)
fn = gfdm_amd.lib().gfdm_hip_receiver_demodulate_host
tm = gfdm_amd.lib().gfdm_hip_host_call_times
def run(nb, reps=2000):
    args = [dem._h, ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(x.ctypes.data), None, ctypes.c_int64(nb)]
    ns = (ctypes.c_int64 * 5)()
    rows, tot = [], []
    for r in range(reps + 50):
        t0 = time.perf_counter_ns(); fn(*args); t1 = time.perf_counter_ns(); tm(ns)
        if r >= 50: rows.append(list(ns)); tot.append(t1 - t0)
    med = np.median(np.array(rows), axis=0) / 1e3
    return "total %.2f us | setup %.2f copy %.2f launch %.2f ticket-post %.2f wait %.2f | outside the pipe (ctypes, guard, return) %.2f" % (np.median(tot) / 1e3, *med, np.median(tot) / 1e3 - med.sum())
for nb in (1, 16, 64):
    print("pageable   %3d blocks:" % nb, run(nb))
with gfdm_amd.registered_host(x, out):
    for nb in (1, 16):
        print("registered %3d blocks:" % nb, run(nb))
