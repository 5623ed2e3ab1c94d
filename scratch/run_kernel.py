"""Launch one receiver/modulator variant repeatedly (for rocprofv3 --pmc / --kernel-trace runs)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("GFDM_PKG") or os.path.join(ROOT, "gr-gfdm_amd", "python"))   # GFDM_PKG: A/B against an older package + library
import numpy as np, torch
import gfdm_amd
from gfdm_amd import synth
from gfdm_amd.filters import get_frequency_domain_filter
path = sys.argv[1]; B = int(sys.argv[2]); reps = int(sys.argv[3]); slots = int(sys.argv[4]) if len(sys.argv) > 4 else 8
K, M, L = (int(x) for x in (sys.argv[5], sys.argv[6], sys.argv[7])) if len(sys.argv) > 7 else (64, 9, 2)
N = K * M
dev = torch.device("cuda:0")
if os.environ.get("GFDM_DFT_MX"): gfdm_amd.set_dft_matrix_cores(int(os.environ["GFDM_DFT_MX"]))   # A/B: 2 = dense matrix-core timeslot transforms also where Rader kernels exist
if os.environ.get("GFDM_MX"): gfdm_amd.set_ic_matrix_cores(int(os.environ["GFDM_MX"]))      # A/B: 0 IC rounds on the vector ALU, 2 matrix cores everywhere
taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
mod = gfdm_amd.Modulator(M, K, L, taps); dem = gfdm_amd.Demodulator(M, K, L, taps)
qpsk = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2)
adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, qpsk)
ins, eqs, outs = [], [], []
for s in range(slots):
    sym = synth.qpsk_symbols(s * B, B, N, dev)
    if path == "modulate":                 # no receiver inputs needed: the trace then holds nothing but the measured modulator launches
        ins.append((sym, None, None)); eqs.append(None); outs.append(torch.empty_like(sym))
        continue
    x = mod.modulate(sym); f = synth.channel_response(s * B, B, N, dev); xe = synth.through_channel(x, f)
    ins.append((sym, x, xe)); eqs.append(f); outs.append(torch.empty_like(x))
if path in ("frames_zf_ic2_est", "estimate_frame"):      # channel estimator, stand-alone / fused in front of ZF + 2 IC + demapper (52 active)
    A = (52 * K) // 64
    smap = np.concatenate((np.arange(1, A // 2 + 1), np.arange(K - A // 2, K)))
    pre = np.tile(np.fft.ifft(np.exp(2j * np.pi * np.random.default_rng(0).random(K))) * np.sqrt(K), 2)
    est = gfdm_amd.ChannelEstimator(M, K, A, True, 1, pre)
    advf = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 2, qpsk)
    advf.configure_frames(N, 0, smap, True)
    advf.set_channel_estimator(est)
    rxp = [torch.tensor(np.tile(pre, (B, 1)), dtype=torch.complex64, device=dev) + 0.05 * torch.randn(B, 2 * K, dtype=torch.complex64, device=dev) for _ in range(slots)]
    souts = [torch.empty(B, A * M, dtype=torch.complex64, device=dev) for _ in range(slots)]
torch.cuda.synchronize()
for r in range(reps):
    s = r % slots
    sym, x, xe = ins[s]
    if path == "modulate": mod.modulate(sym, out=outs[s])
    elif path == "demod_mf": dem.demodulate(x, out=outs[s])
    elif path == "demod_zf": dem.demodulate_equalize(xe, eqs[s], out=outs[s])
    elif path == "demod_mf_ic2": adv.demodulate(x, out=outs[s])
    elif path == "demod_zf_ic2": adv.demodulate_equalize(xe, eqs[s], out=outs[s])
    elif path == "frames_zf_ic2_est": advf.demodulate_estimated(x, rxp[s], out=souts[s])
    elif path == "estimate_frame": est.estimate_frame(rxp[s])
torch.cuda.synchronize()
print("done", path, B, reps)
