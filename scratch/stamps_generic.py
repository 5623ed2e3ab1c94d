"""Diagnostic: per-wavefront phase timestamps of the generic modulate kernel (needs a -DGFDM_STAMPS build: scratch/build_variant.sh stamps -DGFDM_STAMPS).
usage: stamps_generic.py <libgfdm_hip.so> <blocks> K M L [dft matrix-core mode] [modulate | demod_mf]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import numpy as np, torch
import gfdm_amd.capi as capi
capi.LIB_PATH = sys.argv[1]
import gfdm_amd
from gfdm_amd import synth
from gfdm_amd.filters import get_frequency_domain_filter
B = int(sys.argv[2]); K, M, L = (int(v) for v in sys.argv[3:6]); N = K * M
if len(sys.argv) > 6: gfdm_amd.set_dft_matrix_cores(int(sys.argv[6]))
dev = torch.device("cuda:0")
taps = get_frequency_domain_filter("rrc", 0.3, M, K, L)
what = sys.argv[7] if len(sys.argv) > 7 else "modulate"
with gfdm_amd.generic_family_for_testing():
    mod = gfdm_amd.Modulator(M, K, L, taps); dem = gfdm_amd.Demodulator(M, K, L, taps)
run = (lambda d: mod.modulate(d[0], out=d[1])) if what == "modulate" else (lambda d: dem.demodulate(d[0], out=d[1]))
W = 4
stamps = torch.zeros(B * W * 16, dtype=torch.int64, device=dev)
slots = 6
data = [(synth.qpsk_symbols(s * B, B, N, dev), torch.empty(B, N, dtype=torch.complex64, device=dev)) for s in range(slots)]
for s in range(slots - 1): run(data[s])
torch.cuda.synchronize()
setter = ctypes.CDLL(sys.argv[1]).gfdm_debug_set_stamp_buffer_generic
setter.argtypes = [ctypes.c_void_p]
assert setter(ctypes.c_void_p(stamps.data_ptr())) == 0
run(data[-1])
torch.cuda.synchronize()
if what == "modulate": names = ["entry", "block in LDS", "timeslot DFT done", "filter done", "subcarrier FFT done", "twiddle done", "timeslot IDFT + stores done"]
else: names = ["entry", "block in LDS", "timeslot DFT + twiddle done", "subcarrier FFT done", "filter done", "timeslot IDFT + stores done"]
NS = len(names)
t = stamps.cpu().numpy().reshape(B * W, 16)[:, :NS].astype(np.float64) * 0.01      # 100 MHz ticks -> us
t -= t[:, 0].min()
print("generic", what, "K M L =", K, M, L, "B =", B, "mode", sys.argv[6] if len(sys.argv) > 6 else "default")
print("  kernel span: %.2f us" % (t[:, NS - 1].max() - t[:, 0].min()))
for i, n in enumerate(names):
    c = t[:, i]
    print("  %-30s min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f us" % (n, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
d = np.diff(t, axis=1)
for i in range(NS - 1):
    print("  phase %-28s median %6.2f  p90 %6.2f  max %6.2f us" % (names[i] + " ->", np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].max()))
life = t[:, NS - 1] - t[:, 0]
print("  wavefront life: median %.2f  p90 %.2f us;  entries: first wave generation %d of %d wavefronts start within 2 us" % (np.median(life), np.percentile(life, 90), int((t[:, 0] < 2).sum()), B * W))
