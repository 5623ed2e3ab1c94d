"""Time the fused frame receiver (prefix removal + ZF demod + 2 IC + demapper), K=64 M=9, 52 active subcarriers, and the TX->RX chain."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import numpy as np, torch
import gfdm_amd
from gfdm_amd import synth
from gfdm_amd.filters import get_frequency_domain_filter
M, K, A, L, cp, cs = 9, 64, 52, 2, 16, 8
N = M * K; FL = cp + N + cs
taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
smap = np.concatenate((np.arange(1, A // 2 + 1), np.arange(K - A // 2, K)))
qpsk = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2)
dev = torch.device("cuda:0")
tx = gfdm_amd.Transmitter(M, K, A, cp, cs, 0, smap, True, L, taps, np.zeros(0, complex), [0], [np.zeros(0, complex)])
adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 2, qpsk); adv.configure_frames(FL, cp, smap, True)
for B in (4096, 65536):
    slots = 6 if B == 4096 else 2
    data = []
    for s in range(slots):
        sym = synth.qpsk_symbols(s * B, B, A * M, dev); fr = tx.transmit(sym)[0]; feq = synth.channel_response(s * B, B, N, dev)
        data.append((sym, fr, feq, torch.empty(B, A * M, dtype=torch.complex64, device=dev)))
    torch.cuda.synchronize()
    for with_eq in (False, True):
        def run(d): adv.demodulate_frames(d[1], d[2] if with_eq else None, out=d[3])
        for d in data: run(d)
        reps = 30
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        torch.cuda.synchronize()
        for r in range(reps):
            ev[r][0].record(); run(data[r % slots]); ev[r][1].record()
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        bytes_ = B * 8 * (N + (N if with_eq else 0) + A * M)      # block samples (+ f_eq) read, active symbols written
        print("frames %6d  eq=%d  %8.1f us  %6.0f GB/s (%4.1f %% of 8 TB/s)  %.3e frames/s" % (B, with_eq, ms * 1e3, bytes_ / ms / 1e6, bytes_ / ms / 1e6 / 80, B / ms * 1e3))
