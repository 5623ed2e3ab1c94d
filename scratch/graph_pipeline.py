"""Experiment: the 4-stream pipelined headline loop captured into ONE hipGraph (fork / join over the side streams) vs eager launches."""
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
mod = gfdm_amd.Modulator(M, K, L, taps); dem = gfdm_amd.Demodulator(M, K, L, np.conj(taps))
L_ = gfdm_amd.lib()
ns, steps = 36, 200
sym = [synth.qpsk_symbols(s * B, B, N, dev) for s in range(ns)]
fr = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(ns)]
out = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(ns)]
def step(s, stream_ptr):
    assert L_.gfdm_hip_modulator_work_device(mod._h, ctypes.c_void_p(fr[s].data_ptr()), ctypes.c_void_p(sym[s].data_ptr()), ctypes.c_int64(B), ctypes.c_void_p(stream_ptr)) == 0
    assert L_.gfdm_hip_receiver_demodulate_device(dem._h, ctypes.c_void_p(out[s].data_ptr()), ctypes.c_void_p(fr[s].data_ptr()), None, ctypes.c_int64(B), ctypes.c_void_p(stream_ptr)) == 0
for S in (1, 2, 4):
    side = [torch.cuda.Stream(device=dev) for _ in range(S)]
    # eager
    for i in range(40): step(i % ns, side[i % S].cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): step(i % ns, side[(i % ns) % S].cuda_stream)
    torch.cuda.synchronize(); te = time.perf_counter() - t0
    # graph
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(cap):
        g.capture_begin()
        for s_ in side: s_.wait_stream(cap)
        for i in range(steps): step(i % ns, side[(i % ns) % S].cuda_stream)
        for s_ in side: cap.wait_stream(s_)
        g.capture_end()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("streams %d: eager %.1f M blocks/s | one graph of %d steps: %.1f M blocks/s (best of 5: %.1f)" % (S, B * steps / te / 1e6, steps, B * steps / np.median(ts) / 1e6, B * steps / min(ts) / 1e6))
