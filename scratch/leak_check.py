"""Create / use / destroy many handles of every kind; device memory and host RSS must stay flat.  python3 scratch/leak_check.py [rounds]"""
import sys, os, gc, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import numpy as np, torch
import gfdm_amd
from gfdm_amd.filters import get_frequency_domain_filter
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(0)
qpsk = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2)
shapes = [(9, 64, 2), (5, 32, 2), (15, 128, 4), (127, 16, 2), (21, 37, 2)]
def once(i):
    M, K, L = shapes[i % len(shapes)]
    taps = get_frequency_domain_filter("rrc", 0.3, M, K, L)
    mod = gfdm_amd.Modulator(M, K, L, taps); dem = gfdm_amd.Demodulator(M, K, L, taps)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, qpsk)
    d = qpsk[rng.integers(0, 4, (3, M * K))]
    x = mod.modulate(d); y = dem.demodulate(x); z = adv.demodulate(x)
    if i % 4 == 0:                           # a chunked host call (staging sets, second stream, copy pool) and a registered one: HostPipe's resources go with the handle
        big = np.tile(x, (200, 1))
        yb = dem.demodulate(big)
        a = gfdm_amd.aligned_copy(big); o = gfdm_amd.aligned_empty(big.shape)
        with gfdm_amd.registered_host(a, o):
            dem.demodulate(a, out=o)
        assert np.array_equal(yb, o)
    est = gfdm_amd.ChannelEstimator(M, K, (K - 4) & ~1, True, 1, np.tile(np.fft.ifft(np.exp(2j * np.pi * rng.random(K))) * np.sqrt(K), 2)) if K >= 16 else None
    del mod, dem, adv, est
for i in range(10): once(i)
gc.collect(); torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]; rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for i in range(rounds): once(i)
gc.collect(); torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]; rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("handles created and destroyed: %d x 3-4;  device memory free before / after: %.1f / %.1f MiB (delta %.2f MiB);  host max RSS before / after: %.1f / %.1f MiB" %
      (rounds, free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20, rss0 / 1024, rss1 / 1024))
assert free0 - free1 < 64 * 2**20, "device memory leak"
assert rss1 - rss0 < 200 * 1024, "host memory leak"
