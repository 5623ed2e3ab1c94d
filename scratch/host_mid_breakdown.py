"""Mid-size pageable host calls (128 ... 2048 blocks of K=64 M=9): time per call and its breakdown (gfdm_hip_host_call_times), median of many calls."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import gfdm_amd as g
from gfdm_amd.filters import get_frequency_domain_filter
M, K, L = 9, 64, 2; N = M * K
dem = g.Demodulator(M, K, L, get_frequency_domain_filter("rrc", 0.2, M, K, L))
sizes = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "128,256,384,512,768,1024,2048").split(",")]
x = (np.random.default_rng(0).standard_normal((max(sizes), N)) + 0j).astype(np.complex64)
out = np.empty_like(x)
print("build", g.build_id(), "pipeline", g.get_host_pipeline())
for nb in sizes:
    xs, os_ = x[:nb], out[:nb]
    for _ in range(5): dem.demodulate(xs, out=os_)
    ts, parts = [], []
    for _ in range(200):
        t0 = time.perf_counter(); dem.demodulate(xs, out=os_); ts.append(time.perf_counter() - t0)
        st = g.host_call_stats(); parts.append([st["ns"][k] for k in ("setup", "copy", "launch", "post", "wait")])
    med = np.median(np.array(parts), axis=0) / 1e3
    print("%5d blocks: median %.0f us (min %.0f)  %.2f M blocks/s; setup %.0f copy %.0f launch %.0f post %.0f wait %.0f us; chunks %d x %d blocks, pool threads %d" %
          (nb, np.median(ts) * 1e6, min(ts) * 1e6, nb / np.median(ts) / 1e6, *med, st["chunks"], st["chunk_blocks"], st["copy_threads"]))
