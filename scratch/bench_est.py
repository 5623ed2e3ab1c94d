"""Time the preamble channel estimator (estimate_frame) and the estimator -> ZF+IC frame receiver chain."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import numpy as np, torch
import gfdm_amd
from gfdm_amd import synth
from gfdm_amd.filters import get_frequency_domain_filter
dev = torch.device("cuda:0")
qpsk = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2)


def timed(fn, reps=30):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    fn(0); torch.cuda.synchronize()
    for r in range(reps):
        ev[r][0].record(); fn(r); ev[r][1].record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


for (M, K, A) in ((9, 64, 52), (15, 128, 110), (31, 256, 220)):
    N = M * K
    rng = np.random.default_rng(0)
    pre = np.tile(np.fft.ifft(np.exp(2j * np.pi * rng.random(K))) * np.sqrt(K), 2)
    est = gfdm_amd.ChannelEstimator(M, K, A, True, 1, pre)
    for B in (4096, 65536):
        slots = 4 if B == 4096 else 2
        rx = [torch.randn(B, 2 * K, dtype=torch.complex64, device=dev) for _ in range(slots)]
        outs = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(slots)]
        L = gfdm_amd.lib()
        def run(r):
            i = r % slots
            gfdm_amd.capi._check(L.gfdm_hip_channel_estimator_estimate_frame_device(est._h, outs[i].data_ptr(), rx[i].data_ptr(), B, torch.cuda.current_stream().cuda_stream))
        ms = timed(run)
        bytes_ = B * 8 * (2 * K + N)
        print("estimate_frame K=%3d M=%2d frames %6d  %8.1f us  %6.0f GB/s (%4.1f %% of 8 TB/s)  %.3e frames/s" % (K, M, B, ms * 1e3, bytes_ / ms / 1e6, bytes_ / ms / 1e6 / 80, B / ms * 1e3))

# fused estimator + ZF + 2 IC + demapper vs estimate_frame followed by the frame receiver
for (M, K, A, L) in ((9, 64, 52, 2), (15, 128, 110, 4)):
    N = M * K
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    smap = np.concatenate((np.arange(1, A // 2 + 1), np.arange(K - A // 2, K)))
    rng = np.random.default_rng(0)
    pre = np.tile(np.fft.ifft(np.exp(2j * np.pi * rng.random(K))) * np.sqrt(K), 2)
    est = gfdm_amd.ChannelEstimator(M, K, A, True, 1, pre)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 2, qpsk)
    adv.configure_frames(N, 0, smap, True)
    adv.set_channel_estimator(est)
    for B in (4096, 65536):
        slots = 4 if B == 4096 else 2
        blocks = [torch.randn(B, N, dtype=torch.complex64, device=dev) for _ in range(slots)]
        rx = [torch.tensor(np.tile(pre, (B, 1)), dtype=torch.complex64, device=dev) + 0.05 * torch.randn(B, 2 * K, dtype=torch.complex64, device=dev) for _ in range(slots)]
        outs = [torch.empty(B, A * M, dtype=torch.complex64, device=dev) for _ in range(slots)]
        feq = [torch.empty(B, N, dtype=torch.complex64, device=dev) for _ in range(slots)]
        ms_f = timed(lambda r: adv.demodulate_estimated(blocks[r % slots], rx[r % slots], out=outs[r % slots]))
        def chain(r):
            i = r % slots
            gfdm_amd.capi._check(gfdm_amd.lib().gfdm_hip_channel_estimator_estimate_frame_device(est._h, feq[i].data_ptr(), rx[i].data_ptr(), B, torch.cuda.current_stream().cuda_stream))
            adv.demodulate_frames(blocks[i], feq[i], out=outs[i])
        ms_c = timed(chain)
        ms_v = timed(lambda r: adv.demodulate_frames(blocks[r % slots], feq[r % slots], out=outs[r % slots]))
        bf = B * 8 * (N + 2 * K + A * M)
        bv = B * 8 * (2 * N + A * M)
        print("ZF+IC2+demap K=%3d M=%2d frames %6d: fused-estimator %7.1f us (%5.0f GB/s, %4.1f %%) | given f_eq %7.1f us (%5.0f GB/s, %4.1f %%) | estimate_frame + receiver %7.1f us" % (
            K, M, B, ms_f * 1e3, bf / ms_f / 1e6, bf / ms_f / 1e6 / 80, ms_v * 1e3, bv / ms_v / 1e6, bv / ms_v / 1e6 / 80, ms_c * 1e3))
