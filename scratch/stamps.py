"""Diagnostic: per-wave phase timestamps of the row-lane receive kernel (needs the -DGFDM_STAMPS build in /tmp/stamps/libgfdm_hip.so)."""
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
K, M, L = 64, 9, 2; N = K * M
dev = torch.device("cuda:0")
taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
mod = gfdm_amd.Modulator(M, K, L, taps); dem = gfdm_amd.Demodulator(M, K, L, taps)
qpsk = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2)
adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, qpsk)
L_ = gfdm_amd.lib()
stamps = torch.zeros(B * 8, dtype=torch.int64, device=dev)
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
for s in range(slots - 1): run(s)                       # warm-up without stamps
torch.cuda.synchronize()
L_.gfdm_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
assert L_.gfdm_debug_set_stamp_buffer(ctypes.c_void_p(stamps.data_ptr())) == 0
run(slots - 1)
torch.cuda.synchronize()
t = stamps.cpu().numpy().reshape(B, 8)[:, :6].astype(np.float64) * 0.01      # 100 MHz ticks -> us
t -= t[:, 0].min()
names = ["entry", "loads done", "A+B (dft, fft) done", "C+D (eq, filter, idft) done", "IC done", "stores done"]
print(path, "B =", B)
for i, n in enumerate(names):
    c = t[:, i]
    print("  %-28s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (n, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
d = np.diff(t, axis=1)
for i in range(5):
    print("  phase %-26s median %6.2f  p90 %6.2f  max %6.2f us" % (names[i] + " -> next", np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].max()))
