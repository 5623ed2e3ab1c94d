"""Random (timeslots, subcarriers, overlap) shapes, random subcarrier maps, complex taps, every receive mode, frames + demapper,
the fused transmitter and the stand-alone stages, against the float64 oracle.  Not part of the test suite (run-time instantiation of
many shapes takes minutes):   python3 scratch/fuzz_shapes.py [seed] [seconds]
FUZZ_GENERIC=1: every handle on the generic kernel family (up to 300 timeslots, blocks up to 30 000 samples incl. the global-scratch form), the
matrix-core mode of its timeslot transforms (gfdm_hip_set_dft_matrix_cores 0 / 1 / 2) drawn per shape."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gfdm_amd
import gfdm_ref as R
from gfdm_amd.filters import get_frequency_domain_filter

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
rng = np.random.default_rng(seed)
GENERIC = bool(os.environ.get("FUZZ_GENERIC"))
import contextlib
TOL, GUARD = 1e-5, 1e-4


def rel(a, b):
    a = np.asarray(a).reshape(b.shape[0], -1); b = np.asarray(b).reshape(b.shape[0], -1)
    return float(np.max(np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-30)))


def qpsk(shape):
    return ((1 - 2 * rng.integers(0, 2, shape)) + 1j * (1 - 2 * rng.integers(0, 2, shape))) / np.sqrt(2)


t0 = time.time(); n = 0; worst = 0.0; fams = {}
while time.time() - t0 < budget:
    kind = rng.integers(0, 4)
    if kind == 0: K = int(2 ** rng.integers(2, 10))
    elif kind == 1: K = int(rng.choice([6, 10, 12, 14, 15, 18, 20, 24, 30, 36, 40, 48, 60, 72, 80, 96, 100, 112, 120, 144, 160, 192, 208, 224, 240]))
    else: K = int(rng.integers(2, 300))
    M = int(rng.integers(1, 34)) if rng.random() < 0.85 else int(rng.integers(34, 70))
    L = int(rng.choice([2, 2, 2, 3, 4, 5, 6, 8]))
    mxmode = 1
    if GENERIC:
        K = int(rng.integers(2, 70)) if rng.random() < 0.8 else int(rng.integers(70, 400))
        M = int(rng.integers(20, 140)) if rng.random() < 0.8 else int(rng.integers(140, 300))
        mxmode = int(rng.integers(0, 3))
        gfdm_amd.set_dft_matrix_cores(mxmode)
        if L > K or M * K > 30000: continue
    elif L > K or M * K > 12000: continue
    N, B = M * K, int(rng.integers(1, 9))
    alpha = float(rng.choice([0.1, 0.2, 0.35, 0.5, 1.0]))
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    if rng.random() < 0.3:                                              # complex, asymmetric taps: the general IC kernel
        taps = taps * np.exp(1j * rng.uniform(-np.pi, np.pi, L * M)) * rng.uniform(0.5, 1.5, L * M)
    nt = R.normalize_taps(taps, M)
    A = int(rng.integers(1, K + 1))
    smap = np.sort(rng.choice(K, A, replace=False))
    ic_iter = int(rng.integers(0, 4)); pc = int(rng.random() < 0.2)
    tag = "M=%d K=%d L=%d A=%d B=%d ic=%d pc=%d a=%.2f mx=%d" % (M, K, L, A, B, ic_iter, pc, alpha, mxmode)
    try:
        with (gfdm_amd.generic_family_for_testing() if GENERIC else contextlib.nullcontext()):
            mod, dem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
            adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, ic_iter, R.qpsk_points(), do_phase_compensation=pc)
        fams[dem.kernel_name()] = fams.get(dem.kernel_name(), 0) + 1
        d = np.zeros((B, K, M), complex); d[:, smap, :] = qpsk((B, A, M)); d = d.reshape(B, N)
        x = R.modulate(d, nt, M, K, L)
        feq = np.fft.fft(np.array([1, .4 - .2j, .1j][:min(3, N)]), N)[None, :] * np.exp(0.05j * np.arange(B))[:, None]
        xe = np.fft.ifft(np.fft.fft(x, axis=-1) * feq, axis=-1)
        errs = [rel(mod.modulate(d), x), rel(dem.demodulate(x), R.demodulate(x, nt, M, K, L)),
                rel(dem.demodulate_equalize(xe, feq), R.demodulate(xe, nt, M, K, L, feq)),
                rel(dem.fft_filter_downsample(x), R.fft_filter_downsample(x, nt, M, K, L))]
        for inp, eq in ((x, None), (xe, feq)):
            ref, st = R.advanced_receive(inp, nt, M, K, L, smap, R.qpsk_points(), ic_iter, f_eq=eq, kind="qpsk", return_stages=True,
                                         do_phase_compensation=pc)
            keep = np.ones(B, bool)
            for dd in ([st["d0"]] + st["iters"][:-1]) if ic_iter else []:
                v = dd.reshape(-1, K, M)[:, smap, :]
                keep &= np.minimum(np.abs(v.real), np.abs(v.imag)).reshape(B, -1).min(axis=1) > GUARD
            got = adv.demodulate(inp) if eq is None else adv.demodulate_equalize(inp, eq)
            if keep.any(): errs.append(rel(got[keep], ref[keep]) / (5 if pc else 1))
        # frames in, demapped out
        per_ts = bool(rng.integers(0, 2)); off = int(rng.integers(0, 7)); nout = int(rng.integers(1, A * M + 1))
        dem.configure_frames(N + 11, off, smap, per_ts)
        frames = rng.standard_normal((B, N + 11)) + 1j * rng.standard_normal((B, N + 11)); frames[:, off:off + N] = xe
        errs.append(rel(dem.demodulate_frames(frames, feq, noutput_size=nout), R.demap_from_resources(R.demodulate(xe, nt, M, K, L, feq), M, K, smap, per_ts, nout)))
        # fused transmitter + stand-alone stages
        cs = int(rng.integers(0, 6)); cp = int(rng.integers(0, min(N, 9))); shift = int(rng.integers(0, cs + 1)); ramp = int(rng.integers(0, min(3, (N + cp + cs) // 2) + 1))
        if cp + shift > N: shift = 0
        window = rng.standard_normal(N + cp + cs) + 1j * rng.standard_normal(N + cp + cs)
        pre = rng.standard_normal(5) + 1j * rng.standard_normal(5)
        nin = int(rng.integers(1, A * M + 1))
        sym = qpsk((B, nin))
        with (gfdm_amd.generic_family_for_testing() if GENERIC else contextlib.nullcontext()):
            tx = gfdm_amd.Transmitter(M, K, A, cp, cs, ramp, smap, per_ts, L, taps, window, [shift], [pre])
        errs.append(rel(tx.transmit(sym, ninput_size=nin)[0], R.transmit(sym, nt, M, K, L, smap, per_ts, cp, cs, ramp, window, shift, pre)))
        rm = gfdm_amd.ResourceMapper(M, K, A, smap, per_ts)
        grid = rm.map_to_resources(sym, ninput_size=nin)
        assert np.array_equal(np.atleast_2d(grid), R.map_to_resources(sym, M, K, smap, per_ts).astype(np.complex64)), "map"
        assert np.array_equal(np.atleast_2d(rm.demap_from_resources(grid, noutput_size=nin)), sym.astype(np.complex64)), "demap"
        cpx = gfdm_amd.CyclicPrefixer(N, cp, cs, ramp, window, shift)
        errs.append(rel(np.atleast_2d(cpx.add_cyclic_prefix(x.astype(np.complex64))), R.add_cyclic_prefix(x.astype(np.complex64), cp, cs, ramp, window.astype(np.complex64), shift)) * 10)
        e = max(errs); worst = max(worst, e); n += 1
        if e > TOL or not np.isfinite(e): print("FAIL", tag, dem.kernel_name(), ["%.1e" % v for v in errs], flush=True)
    except Exception as ex:  # noqa: BLE001
        print("EXC ", tag, repr(ex)[:300], flush=True)
print("shapes %d  worst relative error %.2e  families %s  (%.0f s)" % (n, worst, fams, time.time() - t0))
