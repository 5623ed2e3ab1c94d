"""Small and mid-size pageable host calls: how many chunks?  For each block count the call is forced into 1 .. 5 chunks (gfdm_hip_set_host_pipeline chunk_bytes)
and timed (median of 300 calls, MF demodulation and ZF + 2 IC at K=64 M=9); `auto` = the library's own plan."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfdm_amd as g
import gfdm_ref as R
from gfdm_amd.filters import get_frequency_domain_filter
M, K, L = 9, 64, 2; N = M * K
taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
dem = g.Demodulator(M, K, L, taps)
adv = g.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points())
sizes = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "8,16,24,32,48,64,96,112,128,192,256,384,512").split(",")]
x = (np.random.default_rng(0).standard_normal((max(sizes), N)) + 0j).astype(np.complex64)
feq = np.ones_like(x)
out = np.empty_like(x)
REG = bool(os.environ.get("SWEEP_REGISTERED"))          # page-aligned, registered buffers: the kernels run on the caller's memory
if REG:
    x, feq, out = g.aligned_copy(x), g.aligned_copy(feq), g.aligned_empty(x.shape)
    for arr in (x, feq, out): g.register_host(arr)
print("build", g.build_id(), "pipeline", g.get_host_pipeline(), "registered buffers" if REG else "pageable buffers")
def med(call, n=200):
    for _ in range(5): call()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
for name, per_block, call in (("demod_mf", 2 * N * 8, lambda nb: dem.demodulate(x[:nb], out=out[:nb])),
                              ("zf_ic2", 3 * N * 8, lambda nb: adv.demodulate_equalize(x[:nb], feq[:nb], out=out[:nb]) if hasattr(adv, "demodulate_equalize") else None)):
    for nb in sizes:
        row = []
        reps = 200 if nb <= 512 else 60
        g.set_host_pipeline(0, 0, 3, 3, 2)
        try:
            row.append("auto %5.1f (%d)" % (med(lambda: call(nb), reps), g.host_call_stats()["chunks"]))
        except Exception as e:
            print(name, "skipped:", e); break
        for nch in ((1, 2, 3, 4, 5) if nb <= 512 else (2, 3, 4, 5, 6, 8, 10, 12, 16)):
            cb = -(-nb // nch)
            if nch > 1 and cb * (nch - 1) >= nb: continue
            g.set_host_pipeline(0, cb * per_block, 3, 3, 2)
            t = med(lambda: call(nb), reps); st = g.host_call_stats()
            row.append("%d: %5.1f" % (st["chunks"], t))
        print("%-8s %4d blocks (%7.1f KiB staged): %s" % (name, nb, nb * per_block / 1024, "   ".join(row)))
g.set_host_pipeline(0, 0, 3, 3, 2)
