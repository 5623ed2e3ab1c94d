"""Exploratory parity sweep over run-time instantiated shapes (every mode against the float64 oracle)."""
import sys, os, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "gr-gfdm_amd/python"): sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch  # noqa
import gfdm_amd, gfdm_ref as R
from gfdm_amd.filters import get_frequency_domain_filter

def rel(a, b):
    a = np.asarray(a).reshape(b.shape); return float(np.max(np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-30)))
def qpsk(rng, shape): return ((1 - 2 * rng.integers(0, 2, shape)) + 1j * (1 - 2 * rng.integers(0, 2, shape))) / np.sqrt(2)

shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
bad = 0
for (M, K, L) in shapes:
    t0 = time.time()
    rng = np.random.default_rng(M * 1000 + K + L)
    taps = get_frequency_domain_filter("rrc", 0.3, M, K, L)
    ctaps = taps * np.exp(1j * rng.uniform(-3, 3, taps.size))       # complex taps: general IC kernel
    N, B = M * K, 21
    smap = np.arange(K) if K < 8 else np.sort(rng.choice(K, size=max(2, (3 * K) // 4), replace=False))
    errs = {}
    for tag, tp in (("real", taps), ("cplx", ctaps)):
        nt = R.normalize_taps(tp, M)
        mod, dem = gfdm_amd.Modulator(M, K, L, tp), gfdm_amd.Demodulator(M, K, L, tp)
        adv = gfdm_amd.AdvancedReceiver(M, K, L, tp, smap, 3, R.qpsk_points())
        names = (mod.kernel_name(), dem.kernel_name(), adv.kernel_name())
        d = np.zeros((B, K, M), complex); d[:, smap, :] = qpsk(rng, (B, len(smap), M)); d = d.reshape(B, N)
        x = R.modulate(d, nt, M, K, L)
        feq = np.fft.fft(np.array([1, .5, .1j, .1 + .05j]), N)[None, :] * np.exp(0.01j * np.arange(B))[:, None]
        xe = np.fft.ifft(np.fft.fft(x, axis=-1) * feq, axis=-1)
        errs[tag + " mod"] = rel(mod.modulate(d), x)
        errs[tag + " fd"] = rel(dem.fft_filter_downsample(x), R.fft_filter_downsample(x, nt, M, K, L))
        errs[tag + " zf"] = rel(dem.demodulate_equalize(xe, feq), R.demodulate(xe, nt, M, K, L, feq))
        for nm, inp, eq in (("mf+ic", x, None), ("zf+ic", xe, feq)):
            ref, st = R.advanced_receive(inp, nt, M, K, L, smap, R.qpsk_points(), 3, f_eq=eq, kind="qpsk", return_stages=True)
            keep = np.ones(B, bool)
            for dd in [st["d0"]] + st["iters"][:-1]:
                v = dd.reshape(-1, K, M)[:, smap, :]
                keep &= np.minimum(np.abs(v.real), np.abs(v.imag)).reshape(B, -1).min(axis=1) > 1e-4
            got = adv.demodulate(inp) if eq is None else adv.demodulate_equalize(inp, eq)
            errs[tag + " " + nm] = rel(got[keep], ref[keep]) if keep.any() else 0.0
        if tag == "real":                                          # estimator fused (part 2) + frames / demap
            A = len(smap) - (len(smap) % 2)
            try:
                sm2 = np.concatenate((np.arange(1, A // 2 + 1), np.arange(K - A // 2, K))) if A >= 2 and A // 2 + 1 <= K - A // 2 else None
            except Exception:
                sm2 = None
            if sm2 is not None and K >= 8:
                pre = np.tile(np.fft.ifft(np.exp(2j * np.pi * rng.random(K))) * np.sqrt(K), 2)
                est = gfdm_amd.ChannelEstimator(M, K, A, True, 1, pre)
                rxp = (np.tile(pre, (B, 1)) * np.exp(0.3j)).astype(np.complex64) + 0.01 * (rng.standard_normal((B, 2 * K)) + 1j * rng.standard_normal((B, 2 * K)))
                fe = R.estimate_frame(rxp, pre.astype(np.complex64), M, K, A, True)
                errs["est"] = rel(est.estimate_frame(rxp), fe)
                dem.set_channel_estimator(est)
                errs["fused"] = rel(dem.demodulate_estimated(x, rxp), R.demodulate(x, nt, M, K, L, fe))
    worst = max(errs.values())
    flag = "" if worst < 1e-5 else "   <<<<<< FAIL"
    bad += worst >= 1e-5
    print("M=%2d K=%3d L=%d %s  worst %.2e  (%s)  %.1fs%s" % (M, K, L, names, worst, max(errs, key=errs.get), time.time() - t0, flag), flush=True)
print("failures:", bad)
