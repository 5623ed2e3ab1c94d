"""Which stage of the cfg5 input preparation / kernel depends on how a batch is split?  (1000 blocks at once vs 8 x 125)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import numpy as np, torch
import gfdm_amd
from gfdm_amd import synth
from gfdm_amd.filters import get_frequency_domain_filter
K, M, L = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 31, 2)
N = K * M; dev = torch.device("cuda:0"); B = 1000
taps = get_frequency_domain_filter("rrc", 0.1, M, K, L)
mod = gfdm_amd.Modulator(M, K, L, taps); dem = gfdm_amd.Demodulator(M, K, L, np.conj(taps))
def gen(start, n):
    sym = synth.qpsk_symbols(start, n, N, dev)
    x = mod.modulate(sym)
    eq = synth.channel_response(start, n, N, dev)
    xe = synth.through_test_channel(x, start)
    y = dem.demodulate_equalize(xe, eq)
    return sym, x, eq, xe, y
whole = gen(0, B)
parts = [gen(s, 125) for s in range(0, B, 125)]
for i, name in enumerate(("symbols", "modulated", "channel_response", "through_test_channel", "zf_demod")):
    cat = torch.cat([p[i] for p in parts])
    same = torch.equal(whole[i].view(torch.float32), cat.view(torch.float32))
    d = (whole[i] - cat).abs().max().item()
    print("%-22s bit-equal %s  max abs diff %.3e" % (name, same, d))
# the kernel alone on identical inputs, split differently
y_whole = dem.demodulate_equalize(whole[3], whole[2])
y_parts = torch.cat([dem.demodulate_equalize(whole[3][s:s + 125].contiguous(), whole[2][s:s + 125].contiguous()) for s in range(0, B, 125)])
print("kernel on identical inputs, 1 x 1000 vs 8 x 125: bit-equal", torch.equal(y_whole.view(torch.float32), y_parts.view(torch.float32)))
