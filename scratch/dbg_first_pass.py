import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gfdm_amd, gfdm_ref as R
from gfdm_amd.filters import get_frequency_domain_filter
M, K, L = 9, 64, 2
N = M * K
taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
dem = gfdm_amd.Demodulator(M, K, L, taps)
rng = np.random.default_rng(0)
x = rng.standard_normal((8, N)) + 1j * rng.standard_normal((8, N))
ref = R.fft_filter_downsample(x, R.normalize_taps(taps, M), M, K, L) if hasattr(R, "fft_filter_downsample") else None
got = dem.fft_filter_downsample(x)
if ref is None:
    print("no oracle fn"); sys.exit()
err = np.abs(got - ref).reshape(8, K, M)
print("max err", err.max(), "ref max", np.abs(ref).max())
print("per column m:", np.round(err.max(axis=(0, 1)), 3))
print("per row k (first block):", np.round(err[0].max(axis=1), 2))
