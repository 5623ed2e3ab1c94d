"""CPU oracle (numpy, float64) for the gr-gfdm sparse-frequency-domain kernels.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (gr-gfdm_amd/, the C-ABI
library, the C++ classes, the pybind11 module) may import or call this file;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and
only as the checker.

It is a restatement of the reference algorithm, stage by stage, vectorised over
a leading batch axis.  Each function cites the reference lines it follows
(paths relative to the gr-gfdm checkout):

  modulator    lib/modulator_kernel_cc.cc:70-141
  receiver     lib/receiver_kernel_cc.cc:56-63, 99-118, 165-192, 211-225, 274-334
  IC receiver  lib/advanced_receiver_kernel_cc.cc:56-123

Pinning (DESIGN.md section 2 keeps the same two lists):

FIXTURE-PINNED -- checked against golden vectors produced in the build container by the reference's own Python model
`pygfdm` (tests/golden/make_golden*.py, tests/test_oracle.py):
  * modulate (any overlap) and demodulate (overlap == 2): gfdm_modulate_block / gfdm_demodulate_block, exactly
    the expectation the reference's C++ tests use (python/qa_python_bindings.py:254-440);
  * the receiver's filter stage at any overlap: pygfdm's overlap-generic model gfdm_demodulate_fft_loop
    (python/pygfdm/gfdm_receiver.py:190-199; fixtures rxl_*.npz), plus the exact transpose identity with the modulator
    (tests/test_oracle.py::test_receiver_is_transpose_of_modulator);
  * ic_filter_taps, cancel_sc_interference and the IC loop (to_td -> 5 x (QPSK decision, cancel against the
    unchanged S, to_td)): gfdm_get_ic_f_taps, gfdm_remove_sc_interference, gfdm_transform_subcarriers_to_tdomain,
    map_qpsk_stream (python/pygfdm/gfdm_receiver.py:91-114, utils.py:80-82; tests/golden/make_golden_ic.py),
    for real RRC taps and for complex asymmetric taps, BASELINE configs 1-5 and the reference's IC test shapes
    (the IC functions do not depend on the overlap once S is given, so overlap 4 is pinned too);
    plus the reference tests' own known answers (genie IC to 1 place, loop-back convergence:
    python/qa_python_bindings.py:410-415, python/qa_advanced_receiver_sb_cc.py:119,172);
  * QPSK decisions (gr::digital::constellation_qpsk, GNU Radio, outside the reference tree) agree with pygfdm's
    map_qpsk_stream away from exact zeros;
  * transmitter chain, mapper / demapper, prefixer, preamble estimator, estimate_snr: tx_*.npz, est_*.npz, snr_*.npz.

RESTATED, READ AGAINST THE SOURCE -- the C++ has no Python counterpart for these and cannot run here (it needs FFTW3f,
VOLK and GNU Radio headers, none installed; it was NOT built against stand-ins), so no reference-generated fixture
exists or can exist.  decide / phase_offset / advanced_receive below restate the lines; each has a known-answer test
derived from its definition:
  * phase compensation, lib/advanced_receiver_kernel_cc.cc:59-71, 78-91 (decision before rotation, the rotation of S
    persists, rotator with increment 1 = one constant phase, sum of arg differences WITHOUT unwrapping):
    tests/test_oracle.py::test_phase_compensation_removes_a_common_phase;
  * partial subcarrier maps, :109-123 (block zeroed, only map entries decided), and the phase mean running over the
    MAP, :82-90 -- a subcarrier listed twice weighs twice and the divisor is map.size() * timeslots:
    tests/test_parity_gpu.py::test_duplicate_subcarrier_map_entry_counts_twice_in_the_phase_mean;
  * the tie rule of constellation_qpsk::decision_maker, zero -> the negative point (from GNU Radio's source; called
    at :119): tests/test_parity_gpu.py::test_all_zero_and_tie_inputs_follow_the_reference_decision_rule.
  Every other constellation: nearest point, first minimum wins -- parity unpinned.

All arithmetic is complex128; callers round to complex64 where they compare
with float32 results.
"""
import numpy as np

__all__ = [
    "normalize_taps", "ic_filter_taps", "modulate", "fft_filter_downsample",
    "transform_subcarriers_to_td", "cancel_sc_interference", "demodulate",
    "qpsk_points", "decide", "advanced_receive", "phase_offset",
]


def _c128(x):
    return np.asarray(x, dtype=np.complex128)


def normalize_taps(taps, timeslots):
    """taps * 1/sqrt(|sum t conj(t)| / M)  -- lib/modulator_kernel_cc.cc:70-85,
    lib/receiver_kernel_cc.cc:99-113.  The reference computes the factor in
    double and stores float32 taps."""
    t = _c128(taps)
    energy = abs(np.sum(t * np.conj(t)))
    return t * (1.0 / np.sqrt(energy / timeslots))


def ic_filter_taps(ntaps, timeslots, overlap):
    """ic[m] = t[m] * t[(L-1)M + m]  -- lib/receiver_kernel_cc.cc:56-63."""
    t = _c128(ntaps)
    return t[0:timeslots] * t[timeslots * (overlap - 1):timeslots * overlap]


def _part_index(K, L):
    """src/target part positions shared by modulator and receiver:
    tap part ((i + L/2) % L), spectrum part ((k + i + K - L/2) % K)
    -- lib/modulator_kernel_cc.cc:118-122, lib/receiver_kernel_cc.cc:175-178."""
    k = np.arange(K)[:, None]
    i = np.arange(L)[None, :]
    tap_part = np.broadcast_to((i + L // 2) % L, (K, L))
    spec_part = (k + i + K - L // 2) % K
    return tap_part, spec_part


def modulate(symbols, ntaps, M, K, L):
    """modulator_kernel_cc::generic_work -- lib/modulator_kernel_cc.cc:98-141.

    symbols: (..., K*M) subcarrier-major [k][m]; ntaps: normalised taps (L*M).
    Returns (..., K*M) time samples.
    """
    d = _c128(symbols)
    batch = d.shape[:-1]
    D = np.fft.fft(d.reshape(batch + (K, M)), axis=-1)          # :109-110
    t = _c128(ntaps).reshape(L, M)
    part_len = min(M * L // 2, M)                                 # :101
    tap_part, spec_part = _part_index(K, L)
    Y = np.zeros(batch + (K, M), dtype=np.complex128)             # :104
    for i in range(L):                                            # spec_part[:, i] is a permutation of k
        filtered = D * t[tap_part[0, i]]                          # :124-127
        Y[..., spec_part[:, i], :part_len] += filtered[..., :part_len]      # :129-132
    x = np.fft.ifft(Y.reshape(batch + (K * M,)), axis=-1)         # :137-140 (ifft = bwd / N)
    return x


def fft_filter_downsample(frame, ntaps, M, K, L, f_eq=None):
    """receiver_kernel_cc::fft_[equalize_]filter_downsample
    -- lib/receiver_kernel_cc.cc:301-320 with :165-192."""
    x = _c128(frame)
    batch = x.shape[:-1]
    X = np.fft.fft(x, axis=-1)                                    # :304-305
    if f_eq is not None:
        X = X / _c128(f_eq)                                       # :315-316
    X = X.reshape(batch + (K, M))
    t = _c128(ntaps).reshape(L, M)
    tap_part, spec_part = _part_index(K, L)
    S = np.zeros(batch + (K, M), dtype=np.complex128)             # :168
    for i in range(L):
        S += t[tap_part[:, i]] * X[..., spec_part[:, i], :]       # :180-188
    return S.reshape(batch + (K * M,))


def transform_subcarriers_to_td(fd, M, K):
    """receiver_kernel_cc::transform_subcarriers_to_td -- :211-225 (ifft = bwd / M)."""
    S = _c128(fd)
    batch = S.shape[:-1]
    return np.fft.ifft(S.reshape(batch + (K, M)), axis=-1).reshape(batch + (K * M,))


def cancel_sc_interference(td, fd, ictaps, M, K):
    """receiver_kernel_cc::cancel_sc_interference -- :274-299."""
    d = _c128(td)
    batch = d.shape[:-1]
    d = d.reshape(batch + (K, M))
    S = _c128(fd).reshape(batch + (K, M))
    neigh = np.roll(d, 1, axis=-2) + np.roll(d, -1, axis=-2)     # prev_sc + next_sc :279-284
    out = S - _c128(ictaps) * np.fft.fft(neigh, axis=-1)          # :285-296
    return out.reshape(batch + (K * M,))


def demodulate(frame, ntaps, M, K, L, f_eq=None):
    """receiver_kernel_cc::generic_work / generic_work_equalize -- :322-334."""
    return transform_subcarriers_to_td(fft_filter_downsample(frame, ntaps, M, K, L, f_eq), M, K)


def qpsk_points():
    """gr::digital::constellation_qpsk (GNU Radio gr-digital, not in the
    reference tree): points in index order --, +-, -+, ++ over sqrt(2);
    decision_maker = 2*(imag > 0) + (real > 0).  Parity unpinned beyond the
    reference's 1-2 decimal-place IC tests."""
    s = np.sqrt(0.5)
    return np.array([-s - 1j * s, s - 1j * s, -s + 1j * s, s + 1j * s], dtype=np.complex128)


def decide(x, points, kind="nearest"):
    """Hard decision.  kind 'qpsk' follows constellation_qpsk::decision_maker
    (sign test, zero maps to the negative point); 'nearest' is the generic
    minimum-Euclidean-distance rule, first minimum wins."""
    x = _c128(x)
    p = _c128(points)
    if kind == "qpsk":
        idx = 2 * (x.imag > 0).astype(np.int64) + (x.real > 0).astype(np.int64)
    elif kind == "bpsk":
        idx = (x.real > 0).astype(np.int64)
    else:
        dist = np.abs(x[..., None] - p) ** 2
        idx = np.argmin(dist, axis=-1)
    return p[idx]


def phase_offset(detected, demod, smap, M, K):
    """advanced_receiver_kernel_cc::calculate_phase_offset -- .cc:78-91:
    mean over active (k, m) of arg(detected) - arg(demod), no unwrapping."""
    batch = detected.shape[:-1]
    a = np.angle(detected.reshape(batch + (K, M))[..., smap, :])
    b = np.angle(demod.reshape(batch + (K, M))[..., smap, :])
    return (a - b).sum(axis=(-1, -2)) / (len(smap) * M)


def advanced_receive(frame, ntaps, M, K, L, smap, points, ic_iter, f_eq=None,
                     do_phase_compensation=0, kind="nearest", return_stages=False):
    """advanced_receiver_kernel_cc::generic_work[_equalize] -- .cc:93-107 with
    perform_ic_iterations :56-76 and map_symbols_to_constellation_points :109-123."""
    smap = np.asarray(smap, dtype=np.int64)
    ic = ic_filter_taps(ntaps, M, L)
    S = fft_filter_downsample(frame, ntaps, M, K, L, f_eq)
    out = transform_subcarriers_to_td(S, M, K)
    batch = out.shape[:-1]
    stages = {"S": S.copy(), "d0": out.copy(), "iters": []}
    for j in range(ic_iter):
        dec = np.zeros(batch + (K, M), dtype=np.complex128)                  # memset :112
        dec[..., smap, :] = decide(out.reshape(batch + (K, M))[..., smap, :], points, kind)
        dec = dec.reshape(batch + (K * M,))
        if do_phase_compensation > 0 and j == 0:                            # :59-71
            phi = phase_offset(dec, out, smap, M, K)
            S = S * np.exp(1j * np.asarray(phi))[..., None]
        fd = cancel_sc_interference(dec, S, ic, M, K)                        # :72-73
        out = transform_subcarriers_to_td(fd, M, K)                          # :74
        stages["iters"].append(out.copy())
    if return_stages:
        stages["dec_margin"] = None
        return out, stages
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Composite transmitter: resource mapper -> modulator -> cyclic prefix + ramp -> preamble

def map_to_resources(symbols, M, K, smap, per_timeslot=True):
    """resource_mapper_kernel_cc::map_to_resources -- lib/resource_mapper_kernel_cc.cc:74-89, 108-134.
    symbols: (..., n) with n <= len(smap)*M; output (..., K*M) subcarrier-major, zero elsewhere.
    The reference walks the SORTED subcarrier map (constructor sorts it, :55)."""
    s = _c128(symbols)
    batch = s.shape[:-1]
    smap = np.sort(np.asarray(smap, dtype=np.int64))
    A = len(smap)
    full = np.zeros(batch + (A * M,), dtype=np.complex128)
    full[..., :s.shape[-1]] = s
    grid = np.zeros(batch + (K, M), dtype=np.complex128)
    if per_timeslot:
        grid[..., smap, :] = np.swapaxes(full.reshape(batch + (M, A)), -1, -2)     # symbol t*A + a -> (smap[a], t)
    else:
        grid[..., smap, :] = full.reshape(batch + (A, M))                          # symbol a*M + t -> (smap[a], t)
    return grid.reshape(batch + (K * M,))


def add_cyclic_prefix(block, cp_len, cs_len, ramp_len, window_taps, cyclic_shift=0):
    """add_cyclic_prefix_cc::add_cyclic_prefix -- lib/add_cyclic_prefix_cc.cc:66-98.  window_taps has either the whole
    window (block+cp+cs) or just 2*ramp_len taps; only its first and last ramp_len entries are used (:51-56)."""
    x = _c128(block)
    N = x.shape[-1]
    w = _c128(window_taps)
    cp_start = N - cp_len - cyclic_shift
    out = np.concatenate((x[..., cp_start:], x, x[..., :cs_len - cyclic_shift]), axis=-1)
    if ramp_len > 0:
        out = out.copy()
        out[..., :ramp_len] *= w[:ramp_len]
        out[..., out.shape[-1] - ramp_len:] *= w[len(w) - ramp_len:]
    return out


def transmit(symbols, ntaps, M, K, L, smap, per_timeslot, cp_len, cs_len, ramp_len, window_taps, cyclic_shift, preamble):
    """transmitter_kernel::modulate + add_frame -- lib/transmitter_kernel.cc:78-107 for one cyclic shift."""
    block = modulate(map_to_resources(symbols, M, K, smap, per_timeslot), ntaps, M, K, L)
    body = add_cyclic_prefix(block, cp_len, cs_len, ramp_len, window_taps, cyclic_shift)
    pre = np.broadcast_to(_c128(preamble), body.shape[:-1] + (len(preamble),))
    return np.concatenate((pre, body), axis=-1)


# ---------------------------------------------------------------------------------------------------------------------
# Receiver-side counterparts: cyclic prefix removal in front, resource demapper behind

def remove_cyclic_prefix(frames, cp_len, block_len):
    """add_cyclic_prefix_cc::remove_cyclic_prefix -- lib/add_cyclic_prefix_cc.cc:100-104."""
    return _c128(frames)[..., cp_len:cp_len + block_len]


def demap_from_resources(grid, M, K, smap, per_timeslot=True, noutput_size=None):
    """resource_mapper_kernel_cc::demap_from_resources -- lib/resource_mapper_kernel_cc.cc:91-106,136-163 (sorted map).
    grid: (..., K*M) subcarrier-major; returns (..., noutput_size) symbols in mapper order.  (The reference's
    per-subcarrier loop writes one element past noutput_size when truncating, :157-161; that overrun is not restated.)"""
    g = _c128(grid)
    batch = g.shape[:-1]
    smap = np.sort(np.asarray(smap, dtype=np.int64))
    act = g.reshape(batch + (K, M))[..., smap, :]                       # (..., A, M)
    out = np.swapaxes(act, -1, -2) if per_timeslot else act             # per timeslot: symbol t*A + a
    out = out.reshape(batch + (len(smap) * M,))
    return out if noutput_size is None else out[..., :noutput_size]


# ---------------------------------------------------------------------------------------------------------------------
# Preamble channel estimator: preamble_channel_estimator_cc, lib/preamble_channel_estimator_cc.cc

def gaussian_taps(n_taps=9, sigma_sq=1.0):
    """initialize_gaussian_filter -- :84-97 (float32 arithmetic in the reference)."""
    i = np.arange(n_taps, dtype=np.float64) - (n_taps // 2)
    t = np.exp(-0.5 * i * i / sigma_sq)
    return t / t.sum()


def estimate_preamble_channel(rx_preamble, preamble, fft_len):
    """estimate_preamble_channel -- :118-145: per half  FFT_K(rx half) * (0.5 / FFT_K(preamble half)), halves summed."""
    rx = _c128(rx_preamble)
    p = _c128(preamble)
    K = fft_len
    with np.errstate(divide="ignore", invalid="ignore"):
        inv0 = 0.5 / np.fft.fft(p[:K])
        inv1 = 0.5 / np.fft.fft(p[K:2 * K])
    return np.fft.fft(rx[..., :K], axis=-1) * inv0 + np.fft.fft(rx[..., K:2 * K], axis=-1) * inv1


def filter_preamble_estimate(estimate, fft_len, active, is_dc_free, taps=None):
    """filter_preamble_estimate -- :147-187: active bins in fftshift order (negative half first), DC bin replaced by the mean
    of its neighbours when dc-free, edges extended by replication, 9-tap Gaussian smoothing."""
    e = _c128(estimate)
    K, A = fft_len, active
    g = gaussian_taps() if taps is None else np.asarray(taps, dtype=np.float64)
    nt = len(g)
    off = 1 if is_dc_free else 0
    parts = [np.repeat(e[..., K - A // 2:K - A // 2 + 1], nt // 2, axis=-1), e[..., K - A // 2:K]]
    if is_dc_free:
        parts.append(((e[..., K - 1] + e[..., 1]) / 2.0)[..., None])
    parts.append(e[..., off:off + A // 2])
    parts.append(np.repeat(e[..., off + A // 2 - 1:off + A // 2], nt // 2, axis=-1))
    inter = np.concatenate(parts, axis=-1)
    n_out = A + off
    out = np.zeros(e.shape[:-1] + (n_out,), dtype=np.complex128)
    for t in range(nt):
        out += inter[..., t:t + n_out] * g[t]
    return out


def interpolate_frame(filtered, timeslots, fft_len, active, is_dc_free):
    """interpolate_frame -- :229-262: linear interpolation of the smoothed K-bin estimate to the M*K bins of a block.
    Without dc-free the reference leaves bins [(A/2 - 1) M, (A/2) M) unwritten; they are filled like the neighbouring
    constant region here (documented deviation, the tests compare only bins the reference writes)."""
    est = _c128(filtered)
    M, K, A = timeslots, fft_len, active
    n_est = A + (1 if is_dc_free else 0)
    N = M * K
    center = N // 2
    dead = K - A
    fe = np.zeros(est.shape[:-1] + (N,), dtype=np.complex128)
    fe[..., M * A // 2 - (0 if is_dc_free else M):center] = est[..., n_est - 1:n_est]
    fe[..., center:center + M * dead // 2] = est[..., 0:1]
    j = np.arange(M) / float(M)
    for i in range(n_est // 2):
        seg = est[..., i:i + 1] + (est[..., i + 1:i + 2] - est[..., i:i + 1]) * j
        s0 = center + M * dead // 2 + i * M
        fe[..., s0:s0 + M] = seg
    for i in range(n_est // 2, n_est - 1):
        seg = est[..., i:i + 1] + (est[..., i + 1:i + 2] - est[..., i:i + 1]) * j
        s0 = (i - n_est // 2) * M
        fe[..., s0:s0 + M] = seg
    return fe


def estimate_frame(rx_preamble, preamble, timeslots, fft_len, active, is_dc_free):
    """estimate_frame -- :272-281."""
    est = estimate_preamble_channel(rx_preamble, preamble, fft_len)
    return interpolate_frame(filter_preamble_estimate(est, fft_len, active, is_dc_free), timeslots, fft_len, active, is_dc_free)


def estimate_snr(rx_preamble, fft_len, active, is_dc_free):
    """estimate_snr -- :189-227: 2K-point FFT of the two-fold repeated preamble; even bins carry symbol + noise energy, odd bins
    noise only.  Returns (snr_lin, cnrs[active])."""
    rx = _c128(rx_preamble)
    K, A = fft_len, active
    S = np.abs(np.fft.fft(rx[..., :2 * K], axis=-1)) ** 2
    half = A // 2
    off = 1 if is_dc_free else 0
    pos_hi = 2 * (np.arange(half) + off)
    pos_lo = 2 * (np.arange(half) + (K - A) // 2 + K // 2)
    se = np.concatenate((S[..., pos_hi], S[..., pos_lo]), axis=-1)
    ne = np.concatenate((S[..., pos_hi + 1], S[..., pos_lo + 1]), axis=-1)
    sym, noise = se.sum(axis=-1), ne.sum(axis=-1)
    with np.errstate(divide="ignore", invalid="ignore"):     # noise-free input: snr = inf, as in the reference's float division
        snr = (sym - noise) / noise
        cnrs = se * np.asarray(snr / (sym / A))[..., None]
    return snr, cnrs
