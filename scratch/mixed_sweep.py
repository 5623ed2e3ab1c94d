"""Row-lane kernels instantiated at run time for subcarrier counts that are not a power of two: parity sweep + timing."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "gr-gfdm_amd", "python"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import torch
import gfdm_amd
import gfdm_ref as R
from gfdm_amd.filters import get_frequency_domain_filter

def rel(a, b):
    a = np.asarray(a).reshape(-1); b = np.asarray(b).reshape(-1)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))

shapes = [(5, 12, 2), (25, 96, 2), (9, 48, 2), (15, 80, 2), (7, 240, 2), (9, 15, 2), (3, 6, 2), (4, 100, 2), (5, 20, 6), (9, 24, 2), (9, 36, 2),
          (9, 60, 2), (9, 72, 4), (9, 112, 2), (9, 120, 2), (9, 144, 2), (9, 160, 2), (9, 192, 2), (9, 224, 2), (16, 3, 2), (9, 10, 2), (32, 14, 2)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
rng = np.random.default_rng(1)
for (M, K, L) in shapes:
    N, B = M * K, 19
    taps = get_frequency_domain_filter("rrc", 0.3, M, K, L)
    nt = R.normalize_taps(taps, M)
    t0 = time.time()
    mod, dem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points())
    tc = time.time() - t0
    sym = ((1 - 2 * rng.integers(0, 2, (B, N))) + 1j * (1 - 2 * rng.integers(0, 2, (B, N)))) / np.sqrt(2)
    x = R.modulate(sym, nt, M, K, L)
    feq = np.fft.fft(np.array([1, .4 - .2j, .1j]), N)[None, :] * np.exp(0.02j * np.arange(B))[:, None]
    xe = np.fft.ifft(np.fft.fft(x, axis=-1) * feq, axis=-1)
    e = [rel(mod.modulate(sym), x), rel(dem.demodulate(x), R.demodulate(x, nt, M, K, L)),
         rel(dem.demodulate_equalize(xe, feq), R.demodulate(xe, nt, M, K, L, feq))]
    ref = R.advanced_receive(xe, nt, M, K, L, np.arange(K), R.qpsk_points(), 2, f_eq=feq, kind="qpsk")
    got = adv.demodulate_equalize(xe, feq)
    bad = np.abs(got - ref).reshape(B, -1).max(axis=1) > 1e-3          # blocks with a flipped borderline decision
    e.append(rel(got[~bad], ref[~bad]) if (~bad).any() else float("nan"))
    # timing: 8192 blocks device resident
    nb = max(256, (1 << 26) // (N * 8))
    dev = torch.device("cuda:0")
    xs = torch.randn(nb, N, dtype=torch.complex64, device=dev)
    fe = torch.randn(nb, N, dtype=torch.complex64, device=dev) + 2
    out = torch.empty_like(xs)
    res = []
    for name, fn in (("mod", lambda: mod.modulate(xs, out=out)), ("mf", lambda: dem.demodulate(xs, out=out)),
                     ("zf_ic2", lambda: adv.demodulate_equalize(xs, fe, out=out))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        byts = (24 if name == "zf_ic2" else 16) * N * nb
        res.append("%s %.0fus %.0f%%" % (name, us, byts / (us * 1e-6) / 8e12 * 100))
    print("M=%d K=%d L=%d [%s] create %.1fs  err mod %.1e mf %.1e zf %.1e zf+ic %.1e (%d guarded)  nb=%d  %s" % (
        M, K, L, dem.kernel_name(), tc, e[0], e[1], e[2], e[3], int(bad.sum()), nb, "  ".join(res)), flush=True)
