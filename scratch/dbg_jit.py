import sys, os, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gfdm_amd
from gfdm_amd.filters import get_frequency_domain_filter
print("lib loaded", flush=True)
M, K, L = int(sys.argv[1]), int(sys.argv[2]), 2
taps = get_frequency_domain_filter("rrc", 0.3, M, K, L)
print("creating demodulator", flush=True)
d = gfdm_amd.Demodulator(M, K, L, taps)
print("kernel", d.kernel_name(), gfdm_amd.lib().gfdm_hip_last_error(), flush=True)
x = (np.random.randn(3, M * K) + 1j * np.random.randn(3, M * K)).astype(np.complex64)
y = d.demodulate(x)
print("demod ok", np.abs(y).max(), flush=True)
a = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2))
print("adv kernel", a.kernel_name(), flush=True)
z = a.demodulate(x)
print("adv ok", np.abs(z).max(), flush=True)
