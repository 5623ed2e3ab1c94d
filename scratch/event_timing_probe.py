"""Which HIP-event timing of a single kernel agrees with rocprofv3's kernel duration?  K=64 M=9, 4096 blocks per launch, ring of 36 slots.
  isolated   synchronize, event, launch, event              (GPU idle before the launch: includes the wake-up / dispatch latency)
  queued     event, launch, event back to back, no sync     (the packets are queued: kernel + in-order dispatch gap)
  pipelined  one pair around N back-to-back launches / N    (consecutive launches overlap head and tail)
Run beside `rocprofv3 --kernel-trace` of scratch/run_kernel.py on the same box (scratch/gpu_r4.sh events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import numpy as np, torch
import gfdm_amd
from gfdm_amd import synth
from gfdm_amd.filters import get_frequency_domain_filter
K, M, L, B, slots, reps = 64, 9, 2, 4096, 36, 400
N = K * M
dev = torch.device("cuda:0")
taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
mod = gfdm_amd.Modulator(M, K, L, taps); dem = gfdm_amd.Demodulator(M, K, L, taps)
adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2))
x, f, o = [], [], []
for s in range(slots):
    fr = mod.modulate(synth.qpsk_symbols(s * B, B, N, dev)); ch = synth.channel_response(s * B, B, N, dev)
    x.append(synth.through_channel(fr, ch)); f.append(ch); o.append(torch.empty_like(fr))
paths = {"demod_mf": lambda s: dem.demodulate(x[s], out=o[s]), "demod_zf_ic2": lambda s: adv.demodulate_equalize(x[s], f[s], out=o[s]),
         "modulate": lambda s: mod.modulate(x[s], out=o[s])}
for name, go in paths.items():
    for s in range(slots): go(s)
    torch.cuda.synchronize()
    iso = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); go(r % slots); e1.record(); e1.synchronize(); iso.append(e0.elapsed_time(e1) * 1e3)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for r in range(reps):
        ev[r][0].record(); go(r % slots); ev[r][1].record()
    torch.cuda.synchronize()
    queued = [a.elapsed_time(b) * 1e3 for a, b in ev]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps): go(r % slots)
    e1.record(); torch.cuda.synchronize()
    print("%-14s isolated median %.2f us (min %.2f) | queued pairs median %.2f us (min %.2f) | pipelined mean %.2f us" %
          (name, np.median(iso), min(iso), np.median(queued), min(queued), e0.elapsed_time(e1) * 1e3 / reps))
