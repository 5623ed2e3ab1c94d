"""One pageable 65 536-block (or given) MF host call a few times, for rocprofv3 --kernel-trace: how do the chunk kernels of the two streams overlap?"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import gfdm_amd as g
from gfdm_amd.filters import get_frequency_domain_filter
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
M, K, L = 9, 64, 2; N = M * K
dem = g.Demodulator(M, K, L, get_frequency_domain_filter("rrc", 0.2, M, K, L))
x = (np.random.default_rng(0).standard_normal((nb, N)) + 0j).astype(np.complex64)
g.set_host_pipeline(0, int(os.environ.get('CHUNK_MIB', '16')) << 20, int(os.environ.get('DEPTH', '3')), int(os.environ.get('THREADS', '5')), streams)
for _ in range(4):
    dem.demodulate(x)
print(g.host_call_stats())
