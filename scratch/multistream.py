import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import numpy as np, torch
import gfdm_amd
from gfdm_amd import synth
from gfdm_amd.filters import get_frequency_domain_filter
K, M, L, B = 64, 9, 2, 4096; N = K * M
dev = torch.device("cuda:0")
taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
L_ = gfdm_amd.lib()
ns = 37
sym = [synth.qpsk_symbols(s * B, B, N, dev) for s in range(ns)]
fr = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(ns)]
out = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(ns)]
for S in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    mods = [gfdm_amd.Modulator(M, K, L, taps) for _ in range(S)]
    dems = [gfdm_amd.Demodulator(M, K, L, taps) for _ in range(S)]
    def step(i):
        s = i % ns; st = i % S; sp = ctypes.c_void_p(streams[st].cuda_stream)
        L_.gfdm_hip_modulator_work_device(mods[st]._h, ctypes.c_void_p(fr[s].data_ptr()), ctypes.c_void_p(sym[s].data_ptr()), ctypes.c_int64(B), sp)
        L_.gfdm_hip_receiver_demodulate_device(dems[st]._h, ctypes.c_void_p(out[s].data_ptr()), ctypes.c_void_p(fr[s].data_ptr()), None, ctypes.c_int64(B), sp)
    for i in range(20): step(i)
    torch.cuda.synchronize()
    steps = 400
    t0 = time.perf_counter()
    for i in range(steps): step(20 + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("streams %d: %.2f us/step, %.3e blocks/s" % (S, dt / steps * 1e6, B * steps / dt))
