"""CPU tests of the oracle itself: both restatements (numpy float64, plain C float32) against the pygfdm golden
vectors and against the known-answer properties the reference's own tests assert."""
import numpy as np
import pytest

import c_oracle
import gfdm_ref as R
from conftest import assert_places, load_golden, rel_err
from gfdm_amd.filters import get_frequency_domain_filter

TOL_F64 = 1e-12     # numpy oracle vs pygfdm (both float64)
TOL_F32 = 1e-5      # north_star tolerance: relative L2 per block, float32 path vs reference


def test_numpy_oracle_matches_pygfdm_modulator(golden):
    g = golden
    nt = R.normalize_taps(g["taps"], g["M"])
    for sym, ref in ((g["symbols"], g["pygfdm_modulate"]), (g["gauss_symbols"], g["pygfdm_modulate_gauss"])):
        assert rel_err(R.modulate(sym, nt, g["M"], g["K"], g["L"]), ref) < TOL_F64


def test_numpy_oracle_matches_pygfdm_receiver(golden):
    g = golden
    if g["L"] != 2:
        pytest.skip("pygfdm's receiver is only valid for overlap 2 (python/pygfdm/gfdm_receiver.py:54,207)")
    nt = R.normalize_taps(g["taps"], g["M"])
    assert rel_err(R.demodulate(g["pygfdm_modulate"], nt, g["M"], g["K"], g["L"]), g["pygfdm_demodulate"]) < TOL_F64
    assert rel_err(R.demodulate(g["gauss_symbols"], nt, g["M"], g["K"], g["L"]), g["pygfdm_demodulate_gauss"]) < TOL_F64


def test_c_oracle_matches_pygfdm(golden):
    g = golden
    o = c_oracle.COracle(g["M"], g["K"], g["L"], g["taps"])
    got = o.modulate(g["symbols"])
    assert rel_err(got, g["pygfdm_modulate"]) < TOL_F32
    assert_places(got, g["pygfdm_modulate"], 5)                  # qa_python_bindings.py:273,294
    assert rel_err(o.modulate(g["gauss_symbols"]), g["pygfdm_modulate_gauss"]) < TOL_F32
    if g["L"] == 2:
        dem = o.demodulate(g["pygfdm_modulate"])
        assert rel_err(dem, g["pygfdm_demodulate"]) < TOL_F32
        assert_places(dem, g["pygfdm_demodulate"], 5)            # qa_python_bindings.py:341,363
        assert rel_err(o.demodulate(g["gauss_symbols"]), g["pygfdm_demodulate_gauss"]) < TOL_F32


def test_c_oracle_matches_numpy_oracle_all_stages(golden):
    g = golden
    M, K, L = g["M"], g["K"], g["L"]
    o = c_oracle.COracle(M, K, L, g["taps"])
    nt = R.normalize_taps(g["taps"], M)
    assert rel_err(o.filter_taps(), nt) < 1e-6
    assert rel_err(o.ic_filter_taps(), R.ic_filter_taps(nt, M, L)) < 1e-6
    x, feq = g["frame_through_channel"], g["f_eq"]
    S = R.fft_filter_downsample(x, nt, M, K, L, feq)
    assert rel_err(o.fft_filter_downsample(x, feq), S) < TOL_F32
    assert rel_err(o.transform_subcarriers_to_td(S), R.transform_subcarriers_to_td(S, M, K)) < TOL_F32
    assert rel_err(o.cancel_sc_interference(g["symbols"], S), R.cancel_sc_interference(g["symbols"], S, R.ic_filter_taps(nt, M, L), M, K)) < TOL_F32
    assert rel_err(o.demodulate(x, feq), R.demodulate(x, nt, M, K, L, feq)) < TOL_F32
    for pc in (0, 1):
        ref = R.advanced_receive(x, nt, M, K, L, g["smap"], R.qpsk_points(), 2, f_eq=feq, kind="qpsk", do_phase_compensation=pc)
        assert rel_err(o.advanced_receive(x, g["smap"], R.qpsk_points(), 2, f_eq=feq, kind="qpsk", do_phase_compensation=pc), ref) < TOL_F32
        assert rel_err(o.advanced_receive(x, g["smap"], R.qpsk_points(), 2, f_eq=feq, kind="nearest", do_phase_compensation=pc), ref) < TOL_F32


def test_equalizer_undoes_channel(golden):
    """qa_python_bindings.py:365-386 generalised: demodulate_equalize(frame * H, H) == demodulate(frame)."""
    g = golden
    nt = R.normalize_taps(g["taps"], g["M"])
    a = R.demodulate(g["frame_through_channel"], nt, g["M"], g["K"], g["L"], g["f_eq"])
    b = R.demodulate(g["pygfdm_modulate"], nt, g["M"], g["K"], g["L"])
    assert rel_err(a, b) < 1e-10


@pytest.mark.parametrize("name", ["cfg4_k128_m15_l4", "ref_m127_k16_l4", "cfg2_k64_m9", "ref_m16_k4"])
def test_receiver_is_transpose_of_modulator(name):
    """The only pin of the receiver at overlap != 2: with A(t) the modulator matrix (pinned by pygfdm at any
    overlap) the receiver matrix is R(t) = (N/M) * A(t)^T (plain transpose), because the fold
    (receiver_kernel_cc.cc:165-192) is the transpose of the scatter (modulator_kernel_cc.cc:116-132) and the
    DFT matrices are symmetric with ifft = conj(fft)/n.  Checked as a bilinear identity on random vectors."""
    g = load_golden(name)
    M, K, L = g["M"], g["K"], g["L"]
    N = M * K
    nt = R.normalize_taps(g["taps"], M)
    rng = np.random.default_rng(7)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    d = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    # R = (1/M) (I x conj F_M) G^T F_N and A = (1/N) conj(F_N) G (I x F_M)  =>  conj(R(t)) = (N/M) A(conj t)^T
    lhs = np.sum(np.conj(R.demodulate(x, nt, M, K, L)) * d)
    rhs = (N / M) * np.sum(np.conj(x) * R.modulate(d, np.conj(nt), M, K, L))
    assert abs(lhs - rhs) / abs(lhs) < 1e-11


@pytest.mark.parametrize("name", __import__("conftest").rx_overlap_golden_names())
def test_receiver_oracles_match_pygfdm_at_any_overlap(name):
    """The receiver's filter stage for overlap 4, 6, 8 (lib/receiver_kernel_cc.cc:165-192) against the reference's overlap-generic Python
    model gfdm_demodulate_fft_loop (python/pygfdm/gfdm_receiver.py:190-199; tests/golden/make_golden_rx_overlap.py) -- the model
    make_golden.py uses is valid for overlap 2 only.  Both oracles, modulated and arbitrary input."""
    from conftest import load_rx_overlap_golden
    g = load_rx_overlap_golden(name)
    M, K, L = g["M"], g["K"], g["L"]
    nt = R.normalize_taps(g["taps"], M)
    co = c_oracle.COracle(M, K, L, g["taps"])
    for x, ref in ((g["frames"], g["pygfdm_demodulate_fft_loop"]), (g["gauss"], g["pygfdm_demodulate_fft_loop_gauss"])):
        assert rel_err(R.demodulate(x, nt, M, K, L), ref) < 1e-12
        assert rel_err(co.demodulate(x), ref) < 1e-6
        # ... and S itself: IDFT_M(S)/M = model output  <=>  S = DFT_M(model output)
        S = R.fft_filter_downsample(x, nt, M, K, L)
        assert rel_err(S, np.fft.fft(ref.reshape(-1, K, M), axis=-1).reshape(ref.shape)) < 1e-12


def test_ic_stage_oracles_match_pygfdm():
    """The IC stage pinned by the reference's Python model (tests/golden/make_golden_ic.py): gfdm_get_ic_f_taps ==
    ic_filter_taps, gfdm_remove_sc_interference == cancel_sc_interference (real RRC taps and complex asymmetric taps), and
    the composed loop to_td -> 5 x (QPSK decision, cancel against the UNCHANGED S, to_td) == advanced_receiver's IC rounds
    (python/pygfdm/gfdm_receiver.py:91-114, utils.py:80-82)."""
    from conftest import ic_golden_names, load_ic_golden
    assert len(ic_golden_names()) >= 8
    for name in ic_golden_names():
        g = load_ic_golden(name)
        M, K, L = g["M"], g["K"], g["L"]
        allk = np.arange(K)
        for taps, ic_ref, cancel_ref in ((g["taps"], g["pygfdm_ic_taps"], g["pygfdm_cancel"]),
                                         (g["ctaps"], g["pygfdm_ic_ctaps"], g["pygfdm_cancel_ctaps"])):
            nt = R.normalize_taps(taps, M)
            o = c_oracle.COracle(M, K, L, taps)
            assert rel_err(R.ic_filter_taps(nt, M, L), ic_ref) < TOL_F64
            assert rel_err(o.ic_filter_taps(), ic_ref) < 1e-6
            assert rel_err(R.cancel_sc_interference(g["td_in"], g["fd_in"], R.ic_filter_taps(nt, M, L), M, K), cancel_ref) < TOL_F64
            assert rel_err(o.cancel_sc_interference(g["td_in"], g["fd_in"]), cancel_ref) < TOL_F32
        nt = R.normalize_taps(g["taps"], M)
        o = c_oracle.COracle(M, K, L, g["taps"])
        assert rel_err(R.fft_filter_downsample(g["frames"], nt, M, K, L), g["S"]) < TOL_F64
        assert rel_err(R.transform_subcarriers_to_td(g["S"], M, K), g["pygfdm_d0"]) < TOL_F64
        rounds = g["pygfdm_ic_iters"]
        for n in range(1, rounds.shape[0] + 1):
            assert rel_err(R.advanced_receive(g["frames"], nt, M, K, L, allk, R.qpsk_points(), n, kind="qpsk"), rounds[n - 1]) < 1e-11
            assert rel_err(o.advanced_receive(g["frames"], allk, R.qpsk_points(), n, kind="qpsk"), rounds[n - 1]) < TOL_F32
            assert rel_err(o.advanced_receive(g["frames"], allk, R.qpsk_points(), n, kind="nearest"), rounds[n - 1]) < TOL_F32


def test_ic_genie_known_answer():
    """qa_python_bindings.py:388-415: after two genie cancellations the demodulated symbols equal the data to 1 place."""
    g = load_golden("ref_m5_k32_a35")
    M, K, L = g["M"], g["K"], g["L"]
    nt = R.normalize_taps(g["taps"], M)
    o = c_oracle.COracle(M, K, L, g["taps"])
    data, frame = g["symbols"], g["pygfdm_modulate"]
    for impl in ("numpy", "c"):
        if impl == "numpy":
            fd = R.fft_filter_downsample(frame, nt, M, K, L)
            res = R.transform_subcarriers_to_td(R.cancel_sc_interference(data, fd, R.ic_filter_taps(nt, M, L), M, K), M, K)
        else:
            fd = o.fft_filter_downsample(frame)
            res = o.transform_subcarriers_to_td(o.cancel_sc_interference(data, fd))
        assert_places(res, data, 1)


@pytest.mark.parametrize("name,ic,places", [("ref_m9_k64_a100", 64, 2), ("ref_m9_k32_act20", 64, 1), ("cfg2_k64_m9", 2, 1)])
def test_ic_loopback_converges(name, ic, places):
    """qa_advanced_receiver_sb_cc.py:84-119 (2 places) and :134-172 (1 place, 20 active subcarriers, data scaled by 2)."""
    g = load_golden(name)
    M, K, L = g["M"], g["K"], g["L"]
    nt = R.normalize_taps(g["taps"], M)
    scale = 2.0 if name == "ref_m9_k32_act20" else 1.0
    data = g["symbols"] * scale
    frame = R.modulate(data, nt, M, K, L)
    o = c_oracle.COracle(M, K, L, g["taps"])
    for res in (R.advanced_receive(frame, nt, M, K, L, g["smap"], R.qpsk_points(), ic, kind="qpsk"),
                o.advanced_receive(frame, g["smap"], R.qpsk_points(), ic, kind="qpsk")):
        res = res.reshape(-1, K, M)[:, g["smap"], :]
        ref = data.reshape(-1, K, M)[:, g["smap"], :]
        if scale == 1.0:
            assert_places(res, ref, places)
        else:   # decisions are unit-energy QPSK, so the cancelled interference is that of unit symbols; signs must match
            assert np.all(np.sign(res.real) == np.sign(ref.real)) and np.all(np.sign(res.imag) == np.sign(ref.imag))


def test_advanced_receiver_with_zero_iterations_is_plain_receiver():
    """qa_advanced_receiver_sb_cc.py:45-82."""
    g = load_golden("ref_m127_k16_l2")
    nt = R.normalize_taps(g["taps"], g["M"])
    a = R.advanced_receive(g["gauss_symbols"], nt, g["M"], g["K"], g["L"], np.arange(g["K"]), R.qpsk_points(), 0)
    assert rel_err(a, g["pygfdm_demodulate_gauss"]) < TOL_F64


def test_tap_normalisation_and_validation():
    M, K, L = 25, 96, 2                                          # qa_python_bindings.py:304-319
    taps = get_frequency_domain_filter("rrc", 0.35, M, K, L)
    o = c_oracle.COracle(M, K, L, taps)
    assert_places(o.filter_taps(), taps, 6)                     # already energy-M normalised -> unchanged
    assert abs(np.sum(np.abs(o.filter_taps()) ** 2) - M) < 1e-4
    o3 = c_oracle.COracle(M, K, L, 3.0 * taps)
    assert rel_err(o3.filter_taps(), taps) < 1e-6
    with pytest.raises(ValueError):
        c_oracle.COracle(M, K, L, taps[:-1])


def overlap_1_by_definition(d, ntaps, M, K):
    """modulator_kernel_cc::generic_work at overlap 1, straight from its lines (lib/modulator_kernel_cc.cc:101,116-132): part_len = M * 1 / 2 (integer),
    src_part_pos = 0, target_part_pos = k * M, so Y[k M + m] = FFT_M(d_k)[m] * taps[m] for m < M // 2, zero above, x = IFFT_N(Y) / N."""
    D = np.fft.fft(np.asarray(d, complex).reshape(-1, K, M), axis=-1)
    Y = np.zeros_like(D)
    Y[..., :M // 2] = D[..., :M // 2] * np.asarray(ntaps, complex)[:M // 2]
    return np.fft.ifft(Y.reshape(-1, K * M), axis=-1)


@pytest.mark.parametrize("M,K", [(9, 64), (7, 12), (8, 4), (21, 37), (127, 16)])
def test_modulator_overlap_1_follows_the_reference_lines(M, K):
    """Overlap 1 cannot be pinned by a fixture: pygfdm's gfdm_modulate_block fails there (python/pygfdm/gfdm_modulation.py:93-98, tail_length = 0: `X[0:0] += X[-0:]`
    does not broadcast -- tried in the build container), and the C++ carries a FIXME for it (lib/modulator_kernel_cc.cc:111-113).  What the C++ lines DO compute is
    written out above; both oracles must equal it."""
    rng = np.random.default_rng(M * 1000 + K)
    taps = get_frequency_domain_filter("rrc", 0.3, M, K, 2)[:M] * np.exp(0.3j * np.arange(M))       # M complex taps
    nt = R.normalize_taps(taps, M)
    d = rng.standard_normal((3, M * K)) + 1j * rng.standard_normal((3, M * K))
    want = overlap_1_by_definition(d, nt, M, K)
    assert rel_err(R.modulate(d, nt, M, K, 1), want) < TOL_F64
    assert rel_err(c_oracle.COracle(M, K, 1, taps).modulate(d), want) < TOL_F32


# ---------------------------------------------------------------- composite transmitter oracles vs pygfdm frames

def test_transmitter_oracles_match_pygfdm():
    """python/qa_transmitter_cc.py:42-55 composition (map -> modulate -> roll -> cyclic starfix -> pinch -> preamble)."""
    from conftest import load_tx_golden, tx_golden_names
    for name in tx_golden_names():
        g = load_tx_golden(name)
        nt = R.normalize_taps(g["taps"], g["M"])
        co = c_oracle.COracleTx(g["M"], g["K"], g["A"], g["cp"], g["cs"], g["ramp"], g["smap"], g["per_timeslot"], g["L"], g["taps"],
                                g["window"], g["shifts"], g["preambles"])
        assert co.n_in == g["A"] * g["M"] and co.n_out == g["pygfdm_frames"].shape[-1]
        for port, s in enumerate(g["shifts"]):
            ref = g["pygfdm_frames"][port]
            got = R.transmit(g["symbols"], nt, g["M"], g["K"], g["L"], g["smap"], g["per_timeslot"], g["cp"], g["cs"], g["ramp"],
                             g["window"], int(s), g["preambles"][port])
            assert rel_err(got, ref) < 1e-7          # pygfdm's mapper rounds the symbols to complex64 (mapping.py:70)
            assert rel_err(co.work(g["symbols"], port), ref) < TOL_F32
            assert_places(co.work(g["symbols"], port), ref, 5)


def test_estimator_oracle_matches_pygfdm_and_known_channel():
    """preamble_channel_estimator_cc restatement against the reference's Python model of it (make_golden_est.py) and against
    the properties python/qa_channel_estimator_cc.py tests: clean preamble -> all ones (6 places), preamble through a
    4-tap channel -> fft(h, M*K) on the active bins (1 place)."""
    from conftest import est_golden_names, load_est_golden
    assert len(est_golden_names()) >= 4
    for name in est_golden_names():
        g = load_est_golden(name)
        M, K, A = g["M"], g["K"], g["A"]
        got = R.estimate_frame(g["rx_preambles"], g["preamble"], M, K, A, True)
        assert np.max(np.abs(got - g["pygfdm_frame_estimates"])) < 1e-12
        assert np.max(np.abs(got[0] - 1.0)) < 1e-6               # qa_channel_estimator_cc.py:84-85
        fh = np.fft.fft(g["channel"], M * K)
        act = M * A // 2                                         # :118-123
        assert np.max(np.abs(got[1][:act] - fh[:act])) < 0.05 and np.max(np.abs(got[1][-act:] - fh[-act:])) < 0.05
        # not dc-free: the bins the reference writes agree with the dc-free estimate away from DC (no golden: the Python model
        # always overwrites DC), the smoothing taps are the normalised Gaussian
        nd = R.estimate_frame(g["rx_preambles"][1], g["preamble"] + 0.05, M, K, A, False)
        assert nd.shape == (M * K,) and np.all(np.isfinite(nd))
    assert abs(R.gaussian_taps().sum() - 1.0) < 1e-15 and np.argmax(R.gaussian_taps()) == 4


@pytest.mark.parametrize("name", __import__("conftest").snr_golden_names())
def test_estimate_snr_oracle_matches_pygfdm_model_and_the_reference_known_answer(name):
    """estimate_snr (lib/preamble_channel_estimator_cc.cc:189-236) against the reference's Python model of it,
    pygfdm.simulation.estimate_snr0, on the vectors of the reference's own test (python/qa_python_bindings.py:492-529: 4 dB, asserted
    within 1 dB) -- tests/golden/make_golden_snr.py."""
    from conftest import load_snr_golden
    g = load_snr_golden(name)
    snr, cnrs = R.estimate_snr(g["rx_preambles"], g["K"], g["A"], True)
    assert np.max(np.abs(snr / g["pygfdm_estimate_snr0"] - 1.0)) < 1e-6
    limit = 1.0 if g["K"] >= 1024 else 2.0           # the reference's bound holds for its 936 active bins; fewer bins scatter more
    assert np.max(np.abs(10 * np.log10(snr) - g["snr_db"])) < limit
    assert np.allclose(cnrs.sum(axis=-1), g["A"] * snr, rtol=1e-9)


def phase_case(M, K, L, alpha, B=6):
    """clean modulated QPSK blocks on all but the DC subcarrier + the blocks on which a rotation by |phi0| <= 0.06 moves no
    demodulated symbol across the branch cut of arg() (the reference sums arg differences without unwrapping) or across a decision
    boundary (the matched filter leaves self-interference of up to 0.46 on a symbol of modulus 1)"""
    rng = np.random.default_rng(M + K)
    nt = R.normalize_taps(get_frequency_domain_filter("rrc", alpha, M, K, L), M)
    N = M * K
    smap = np.concatenate((np.arange(1, K // 2), np.arange(K // 2 + 1, K)))
    d = np.zeros((B, K, M), complex)
    d[:, smap, :] = ((1 - 2 * rng.integers(0, 2, (B, len(smap), M))) + 1j * (1 - 2 * rng.integers(0, 2, (B, len(smap), M)))) / np.sqrt(2)
    x = R.modulate(d.reshape(B, N), nt, M, K, L)
    d0 = R.demodulate(x, nt, M, K, L).reshape(B, K, M)[:, smap, :]
    keep = np.abs(np.angle(d0)).reshape(B, -1).max(axis=1) + 0.06 < np.pi
    keep &= np.minimum(np.abs(d0.real), np.abs(d0.imag)).reshape(B, -1).min(axis=1) > 0.06 * np.abs(d0).max()
    return nt, smap, x, keep


def test_phase_compensation_removes_a_common_phase():
    """Known answer for lib/advanced_receiver_kernel_cc.cc:59-71,78-91, which has no Python model and no reference test: the first IC
    round measures phi = mean over the active symbols of arg(decision) - arg(demodulated) and rotates S by it.  For a frame rotated by
    phi0 (small enough to leave the decisions and the branch of arg() alone) every demodulated symbol turns by phi0, so
    phi(rotated) = phi(clean) - phi0 EXACTLY and the rotated S lands on S exp(j phi(clean)): the receiver with phase compensation is
    invariant to a common phase of its input, while without it the output keeps carrying phi0."""
    for (M, K, L, alpha) in ((9, 64, 2, 0.2), (15, 128, 4, 0.2), (5, 32, 2, 0.5)):
        nt, smap, x, keep = phase_case(M, K, L, alpha)
        B, N = x.shape
        assert keep.sum() >= 2
        for phi0 in (0.05, -0.03):
            xr = x * np.exp(1j * phi0)
            clean = R.advanced_receive(x, nt, M, K, L, smap, R.qpsk_points(), 2, do_phase_compensation=1, kind="qpsk")
            rot = R.advanced_receive(xr, nt, M, K, L, smap, R.qpsk_points(), 2, do_phase_compensation=1, kind="qpsk")
            assert rel_err(rot[keep], clean[keep]) < 1e-12
            plain = R.advanced_receive(xr, nt, M, K, L, smap, R.qpsk_points(), 2, kind="qpsk")
            nopc = R.advanced_receive(x, nt, M, K, L, smap, R.qpsk_points(), 2, kind="qpsk")
            act = lambda v: v.reshape(B, K, M)[:, smap, :]
            assert abs(np.angle(np.sum(act(plain) * np.conj(act(nopc)))) - phi0) < 0.01     # without compensation phi0 stays in the output
            # and the measured offset itself: phi(rotated) = phi(clean) - phi0
            st = R.advanced_receive(x, nt, M, K, L, smap, R.qpsk_points(), 0, return_stages=True)[1]
            dec = np.zeros((B, K, M), complex)
            dec[:, smap, :] = R.decide(st["d0"].reshape(B, K, M)[:, smap, :], R.qpsk_points(), "qpsk")
            p_clean = R.phase_offset(dec.reshape(B, N), st["d0"], smap, M, K)
            p_rot = R.phase_offset(dec.reshape(B, N), st["d0"] * np.exp(1j * phi0), smap, M, K)
            assert np.max(np.abs(p_rot - (p_clean - phi0))[keep]) < 1e-12


def test_estimator_snr_properties():
    """estimate_snr (lib/preamble_channel_estimator_cc.cc:189-227): noise-free two-fold repetition -> odd bins empty;
    known noise level recovered; cnrs sum to A * snr."""
    from conftest import load_est_golden
    g = load_est_golden("est_cfg2_m9_k64_a52")
    K, A = g["K"], g["A"]
    rng = np.random.default_rng(0)
    clean = g["rx_preambles"][1]
    snrs = []
    for _ in range(200):
        sigma = np.sqrt(np.mean(np.abs(clean) ** 2) / 100.0 / 2)                   # 20 dB over the full band
        snr, cnrs = R.estimate_snr(clean + sigma * (rng.standard_normal(2 * K) + 1j * rng.standard_normal(2 * K)), K, A, True)
        assert cnrs.shape == (A,) and abs(cnrs.sum() - A * snr) < 1e-9 * A * snr
        snrs.append(snr)
    # signal occupies A of K bins: in-band SNR = 100 * K / A; (sym - noise) / noise  => 2x because the even bins collect both halves
    expect = 2 * 100.0 * K / A
    assert abs(np.mean(snrs) / expect - 1.0) < 0.1
    big, _ = R.estimate_snr(clean, K, A, True)
    assert big > 1e6                                                               # window-free circular preamble: (almost) no odd-bin energy


def _pygfdm_grid(g):
    """pygfdm's mapper builds ceil(n / A) timeslots for n symbols (mapping.py:64-66), the C++ mapper always `timeslots`: pad"""
    M, K = g["M"], g["K"]
    grid = g["pygfdm_grid"].reshape(g["pygfdm_grid"].shape[0], K, -1)
    out = np.zeros((grid.shape[0], K, M), complex)
    out[:, :, :grid.shape[2]] = grid
    return out.reshape(-1, K * M)


def test_mapper_and_prefixer_oracles_match_pygfdm():
    """The stand-alone stages (resource_mapper_kernel_cc, add_cyclic_prefix_cc restatements) against the reference's Python model:
    map_to_waveform_resources, demap_from_waveform_resource_grid (mapping.py:53-76) and roll + add_cyclic_starfix + pinch_block
    (cyclic_prefix.py, as composed in qa_transmitter_cc.py:50-53)."""
    from conftest import load_tx_golden, tx_golden_names
    for name in tx_golden_names():
        g = load_tx_golden(name)
        M, K, plen = g["M"], g["K"], g["preambles"].shape[-1]
        assert np.array_equal(R.map_to_resources(g["symbols"], M, K, g["smap"], g["per_timeslot"]).astype(np.complex64), _pygfdm_grid(g).astype(np.complex64))
        # pygfdm demaps in per-timeslot order only (mapping.py:58-61)
        assert np.array_equal(R.demap_from_resources(g["grid_in"], M, K, g["smap"], True), g["pygfdm_demapped"])
        for port, s in enumerate(g["shifts"]):
            got = R.add_cyclic_prefix(g["pygfdm_blocks"], g["cp"], g["cs"], g["ramp"], g["window"], int(s))
            assert rel_err(got, g["pygfdm_frames"][port][:, plen:]) < 1e-12
            # behind the shifted prefix sits the block itself (where the window ramps have not touched it)
            back = R.remove_cyclic_prefix(got, g["cp"] + int(s), M * K)
            clean = slice(max(0, g["ramp"] - g["cp"] - int(s)), M * K - max(0, g["ramp"] - (g["cs"] - int(s))))
            assert np.array_equal(back[:, clean], g["pygfdm_blocks"][:, clean])
