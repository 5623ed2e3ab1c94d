"""Time the fused transmitter kernel (mapper + modulator + cyclic prefix/ramp + preamble), K=64 M=9 52 active subcarriers."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import numpy as np, torch
import gfdm_amd
from gfdm_amd import synth
from gfdm_amd.filters import get_frequency_domain_filter
M, K, A, L, cp, cs, ramp, plen = 9, 64, 52, 2, 16, 8, 8, 160
taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
smap = np.concatenate((np.arange(1, A // 2 + 1), np.arange(K - A // 2, K)))
N = M * K; FL = N + cp + cs
window = np.ones(FL, complex); r = np.arange(ramp) / ramp
window[:ramp] = 0.5 * (1 - np.cos(np.pi * r)); window[-ramp:] = window[:ramp][::-1]
rng = np.random.default_rng(0)
dev = torch.device("cuda:0")
for shifts in ([0], [0, 3, 7, 8]):
    pre = [rng.standard_normal(plen) + 1j * rng.standard_normal(plen) for _ in shifts]
    tx = gfdm_amd.Transmitter(M, K, A, cp, cs, ramp, smap, True, L, taps, window, shifts, pre)
    for B in (4096, 65536):
        slots = 6 if B == 4096 else 2
        syms = [synth.qpsk_symbols(s * B, B, A * M, dev) for s in range(slots)]
        import ctypes
        L_ = gfdm_amd.lib()
        outs = [[torch.empty(B, tx.output_vector_size(), dtype=torch.complex64, device=dev) for _ in shifts] for _ in range(slots)]
        arrs = [(ctypes.c_void_p * len(shifts))(*[o.data_ptr() for o in outs[s]]) for s in range(slots)]
        st = torch.cuda.current_stream().cuda_stream
        def run(s):
            assert L_.gfdm_hip_transmitter_work_device(tx._h, arrs[s], len(shifts), ctypes.c_void_p(syms[s].data_ptr()), A * M, B, ctypes.c_void_p(st)) == 0
        for s in range(slots): run(s)
        reps = 30
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        torch.cuda.synchronize()
        for r_ in range(reps):
            ev[r_][0].record(); run(r_ % slots); ev[r_][1].record()
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        bytes_ = B * 8 * (A * M + len(shifts) * tx.output_vector_size())
        print("ports %d  frames %6d  %8.1f us  %6.0f GB/s (%4.1f %% of 8 TB/s)  %.3e frames/s  [%s]" % (len(shifts), B, ms * 1e3, bytes_ / ms / 1e6, bytes_ / ms / 1e6 / 80, B / ms * 1e3, tx.kernel_name()))
