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
taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
mod = gfdm_amd.Modulator(M, K, L, taps); dem = gfdm_amd.Demodulator(M, K, L, taps)
qpsk = np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2)
adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, qpsk)
ins, eqs, outs = [], [], []
for s in range(slots):
    sym = synth.qpsk_symbols(s * B, B, N, dev)
    x = mod.modulate(sym); f = synth.channel_response(s * B, B, N, dev); xe = synth.through_channel(x, f)
    ins.append((sym, x, xe)); eqs.append(f); outs.append(torch.empty_like(x))
torch.cuda.synchronize()
for r in range(reps):
    s = r % slots
    sym, x, xe = ins[s]
    if path == "modulate": mod.modulate(sym, out=outs[s])
    elif path == "demod_mf": dem.demodulate(x, out=outs[s])
    elif path == "demod_zf": dem.demodulate_equalize(xe, eqs[s], out=outs[s])
    elif path == "demod_mf_ic2": adv.demodulate(x, out=outs[s])
    elif path == "demod_zf_ic2": adv.demodulate_equalize(xe, eqs[s], out=outs[s])
torch.cuda.synchronize()
print("done", path, B, reps)
