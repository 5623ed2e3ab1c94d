"""Event timing variants of the MF demodulate launch (K=64 M=9, 4096 blocks): pair per launch vs one pair around a run of launches."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import numpy as np, torch, ctypes
import gfdm_amd
from gfdm_amd import synth
from gfdm_amd.filters import get_frequency_domain_filter
M, K, L, B = 9, 64, 2, 4096
N = M * K
dev = torch.device("cuda:0")
taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
dem = gfdm_amd.Demodulator(M, K, L, taps)
Lb = gfdm_amd.lib()
ns = 36
x = [torch.randn(B, N, dtype=torch.complex64, device=dev) for _ in range(ns)]
o = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(ns)]
st = torch.cuda.current_stream().cuda_stream
def launch(i):
    Lb.gfdm_hip_receiver_demodulate_device(dem._h, ctypes.c_void_p(o[i % ns].data_ptr()), ctypes.c_void_p(x[i % ns].data_ptr()), None, ctypes.c_int64(B), ctypes.c_void_p(st))
for i in range(50): launch(i)
torch.cuda.synchronize()
for run in (1, 2, 4, 8, 25, 50, 200):
    reps = max(1, 200 // run)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    k = 0
    for r in range(reps):
        ev[r][0].record()
        for _ in range(run):
            launch(k); k += 1
        ev[r][1].record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ev]) / run * 1e3
    print("run of %3d launches per event pair: mean %.2f us  median %.2f us  min %.2f us per launch" % (run, t.mean(), np.median(t), t.min()))
