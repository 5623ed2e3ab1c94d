"""Diagnostic: per-wave phase timestamps of the row-lane receive kernel (needs a -DGFDM_STAMPS build: scratch/build_variant.sh stamps -DGFDM_STAMPS).
usage: stamps.py <libgfdm_hip.so> <demod_mf | demod_zf | demod_mf_ic2 | demod_zf_ic2> <blocks> [K M L]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import numpy as np, torch
import gfdm_amd.capi as capi
capi.LIB_PATH = sys.argv[1]
import gfdm_amd
from gfdm_amd import synth
from gfdm_amd.filters import get_frequency_domain_filter
path, B = sys.argv[2], int(sys.argv[3])
K, M, L = (int(v) for v in sys.argv[4:7]) if len(sys.argv) > 6 else (64, 9, 2); N = K * M
dev = torch.device("cuda:0")
taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
mod = gfdm_amd.Modulator(M, K, L, taps); dem = gfdm_amd.Demodulator(M, K, L, taps)
qpsk = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2)
adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, qpsk)
L_ = gfdm_amd.lib()
W = max(1, K // 64)                                       # waves per block
stamps = torch.zeros(B * W * 8, dtype=torch.int64, device=dev)
slots = 6
data = []
for s in range(slots):
    sym = synth.qpsk_symbols(s * B, B, N, dev); x = mod.modulate(sym); f = synth.channel_response(s * B, B, N, dev)
    data.append((x, synth.through_channel(x, f), f, torch.empty_like(x)))
torch.cuda.synchronize()
def run(s):
    x, xe, f, o = data[s]
    if path == "demod_mf": dem.demodulate(x, out=o)
    elif path == "demod_zf": dem.demodulate_equalize(xe, f, out=o)
    elif path == "demod_zf_ic2": adv.demodulate_equalize(xe, f, out=o)
    elif path == "demod_mf_ic2": adv.demodulate(x, out=o)
for s in range(slots - 1): run(s)                       # warm-up without stamps
torch.cuda.synchronize()
mx = gfdm_amd.capi.lib().gfdm_hip_set_ic_matrix_cores(1); gfdm_amd.capi.lib().gfdm_hip_set_ic_matrix_cores(mx)
part = 0 if "ic" not in path else (4 if (mx == 2 or (mx == 1 and K >= 128)) and 4 <= M <= 16 else 1)
setter = getattr(ctypes.CDLL(sys.argv[1]), "gfdm_debug_set_stamp_buffer_%d_%d_%d_p%d" % (K, M, L, part))
setter.argtypes = [ctypes.c_void_p]
assert setter(ctypes.c_void_p(stamps.data_ptr())) == 0
run(slots - 1)
torch.cuda.synchronize()
t = stamps.cpu().numpy().reshape(B * W, 8)[:, :6].astype(np.float64) * 0.01      # 100 MHz ticks -> us
t -= t[:, 0].min()
names = ["entry", "loads done", "A+B (dft, fft) done", "C+D (eq, filter, idft) done", "IC done", "stores done"]
print(path, "K M L =", K, M, L, "B =", B, "part", part)
print("  kernel span (first entry -> last store): %.2f us" % (t[:, 5].max() - t[:, 0].min()))
for i, n in enumerate(names):
    c = t[:, i]
    print("  %-28s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (n, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
d = np.diff(t, axis=1)
for i in range(5):
    print("  phase %-26s median %6.2f  p90 %6.2f  max %6.2f us" % (names[i] + " -> next", np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].max()))
