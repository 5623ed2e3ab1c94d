"""Pageable host calls with the automatic chunk plan: kernel streams (1, 2) x copy threads (0, 2, 3, 5) x staging sets (2, 3), K=64 M=9 MF demodulation, median us per call."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import gfdm_amd as g
from gfdm_amd.filters import get_frequency_domain_filter
M, K, L = 9, 64, 2; N = M * K
dem = g.Demodulator(M, K, L, get_frequency_domain_filter("rrc", 0.2, M, K, L))
sizes = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "128,256,512,1024,4096,32768").split(",")]
x = (np.random.default_rng(0).standard_normal((max(sizes), N)) + 0j).astype(np.complex64)
out = np.empty_like(x)
print("build", g.build_id())
for nb in sizes:
    reps = 150 if nb <= 1024 else 40 if nb <= 4096 else 8
    cells = []
    for depth in (2, 3):
        for streams in (1, 2):
            for thr in (0, 2, 3, 5):
                g.set_host_pipeline(0, 0, depth, thr, streams)
                for _ in range(3): dem.demodulate(x[:nb], out=out[:nb])
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter(); dem.demodulate(x[:nb], out=out[:nb]); ts.append(time.perf_counter() - t0)
                cells.append("d%d s%d t%d %7.1f" % (depth, streams, thr, np.median(ts) * 1e6))
    print("%6d blocks: %s" % (nb, " | ".join(cells)))
g.set_host_pipeline(0, 0, 3, 3, 2)
