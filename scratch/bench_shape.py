"""Time every path for one (K, M, L) shape: event-bracketed kernel duration over a ring of buffers."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import numpy as np, torch
import gfdm_amd
from gfdm_amd import synth
from gfdm_amd.filters import get_frequency_domain_filter
K, M, L, B = (int(x) for x in sys.argv[1:5]); alpha = float(sys.argv[5]) if len(sys.argv) > 5 else 0.2
N = K * M; dev = torch.device("cuda:0")
taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
if "GFDM_DFT_MX" in os.environ: gfdm_amd.set_dft_matrix_cores(int(os.environ["GFDM_DFT_MX"]))      # generic family: timeslot transforms on the matrix cores (default) / vector ALU
mod = gfdm_amd.Modulator(M, K, L, taps); dem = gfdm_amd.Demodulator(M, K, L, taps)
qpsk = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2)
adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, qpsk)
slots = max(2, min(16, int(3e9 // (B * N * 8 * 4))))
data = []
for s in range(slots):
    sym = synth.qpsk_symbols(s * B, B, N, dev); x = mod.modulate(sym); f = synth.channel_response(s * B, B, N, dev)
    data.append((sym, x, synth.through_channel(x, f), f, torch.empty_like(x)))
torch.cuda.synchronize()
paths = {"modulate": (16, lambda d: mod.modulate(d[0], out=d[4])), "demod_mf": (16, lambda d: dem.demodulate(d[1], out=d[4])),
         "demod_zf": (24, lambda d: dem.demodulate_equalize(d[2], d[3], out=d[4])), "demod_mf_ic2": (16, lambda d: adv.demodulate(d[1], out=d[4])),
         "demod_zf_ic2": (24, lambda d: adv.demodulate_equalize(d[2], d[3], out=d[4]))}
print("K=%d M=%d L=%d B=%d kernel=%s slots=%d" % (K, M, L, B, dem.kernel_name(), slots))
for name, (bps, fn) in paths.items():
    reps = 30
    for r in range(3): fn(data[r % slots])
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for r in range(reps):
        ev[r][0].record(); fn(data[r % slots]); ev[r][1].record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    print("  %-13s %9.1f us  %7.0f GB/s  %5.1f %% of 8 TB/s   %.3e blocks/s" % (name, ms * 1e3, bps * N * B / ms / 1e6, bps * N * B / ms / 1e6 / 80, B / ms * 1e3))
