"""GPU parity tests of the preamble channel estimator (preamble_channel_estimator_cc).
Expectations: the reference's Python model of it (tests/golden/est_*.npz, make_golden_est.py), the numpy oracle stage by
stage, and the properties python/qa_channel_estimator_cc.py checks."""
import numpy as np
import pytest

import gfdm_ref as R
from conftest import assert_places, check_err, est_golden_names, have_gpu, load_est_golden, rel_err
from gfdm_amd.filters import get_frequency_domain_filter

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    if not have_gpu():
        pytest.fail("no MI355X visible: the HIP path cannot run (there is no CPU fallback to test instead)")


def _active_bins(K, A, dc_free):
    off = 1 if dc_free else 0
    return np.concatenate((np.arange(off, off + A // 2), np.arange(K - A // 2, K)))


@pytest.mark.parametrize("name", est_golden_names())
def test_estimate_frame_matches_pygfdm(name):
    import gfdm_amd
    g = load_est_golden(name)
    M, K, A = g["M"], g["K"], g["A"]
    est = gfdm_amd.ChannelEstimator(M, K, A, True, 1, g["preamble"])
    assert (est.timeslots(), est.fft_len(), est.active_subcarriers(), est.frame_len(), est.is_dc_free()) == (M, K, A, M * K, True)
    # compiled shapes, a shape instantiated at run time, and (below) the generic family on the same inputs
    assert est.kernel_name() == ("rowlane" if (K, M) in ((64, 9), (128, 15), (64, 5)) else "rowlane_jit" if (K, M) == (32, 3) else "generic_lds")
    with gfdm_amd.generic_family_for_testing():
        gen = gfdm_amd.ChannelEstimator(M, K, A, True, 1, g["preamble"])
    assert gen.kernel_name() == "generic_lds" and rel_err(gen.estimate_frame(g["rx_preambles"]), g["pygfdm_frame_estimates"]) < TOL
    got = est.estimate_frame(g["rx_preambles"])
    assert got.shape == g["pygfdm_frame_estimates"].shape
    assert rel_err(got, g["pygfdm_frame_estimates"]) < TOL
    assert_places(got, g["pygfdm_frame_estimates"], 4)
    assert np.array_equal(est.estimate_frame(g["rx_preambles"][2]), got[2])       # one preamble == row of the batch
    # qa_channel_estimator_cc.py:84-85 (clean preamble -> ones, 6 places; float32 here: 5) and :118-125 (channel, 1 place)
    assert np.max(np.abs(got[0] - 1.0)) < 1e-5
    fh = np.fft.fft(g["channel"], M * K)
    act = M * A // 2
    assert_places(got[1][:act], fh[:act], 1)
    assert_places(got[1][-act:], fh[-act:], 1)
    assert np.allclose(est.preamble_filter_taps(), R.gaussian_taps(), atol=1e-7)


@pytest.mark.parametrize("M,K,A,dc_free", [(9, 64, 52, True), (9, 64, 52, False), (5, 32, 32, False), (15, 128, 110, True), (7, 12, 8, True),
                                           (3, 48, 40, False), (2, 1024, 936, True)])
def test_estimator_stages_against_oracle(M, K, A, dc_free):
    """every public stage (estimate_preamble_channel, filter_preamble_estimate, interpolate_frame, prepare_for_zf) and their
    fusion, power-of-two and other fft_len, dc-free or not, random full-band preamble (every bin invertible)."""
    import gfdm_amd
    rng = np.random.default_rng(M * K + A + dc_free)
    B = 6
    pre = (rng.standard_normal(2 * K) + 1j * rng.standard_normal(2 * K)) / np.sqrt(2)
    rx = rng.standard_normal((B, 2 * K)) + 1j * rng.standard_normal((B, 2 * K))
    est = gfdm_amd.ChannelEstimator(M, K, A, dc_free, 0, pre)
    assert est.filtered_len() == A + dc_free
    ref_e = R.estimate_preamble_channel(rx, pre, K)
    ref_f = R.filter_preamble_estimate(ref_e, K, A, dc_free)
    ref_i = R.interpolate_frame(ref_f, M, K, A, dc_free)
    got_e = est.estimate_preamble_channel(rx)
    assert rel_err(got_e, ref_e) < TOL
    assert rel_err(est.filter_preamble_estimate(ref_e), ref_f) < TOL
    assert rel_err(est.interpolate_frame(ref_f), ref_i) < TOL
    fused = est.estimate_frame(rx)
    assert rel_err(fused, ref_i) < TOL
    chained = est.interpolate_frame(est.filter_preamble_estimate(got_e))
    assert rel_err(chained, fused) < 1e-6                                         # same arithmetic up to float32 round trips
    assert rel_err(est.prepare_for_zf(fused), np.conj(1.0 / fused.astype(np.complex128))) < TOL      # same input: 1/x is ill-conditioned near fades
    if not dc_free:                                                               # bins the reference itself writes
        written = np.ones(M * K, bool)
        written[(A // 2 - 1) * M:(A // 2) * M] = False
        assert rel_err(fused[:, written], ref_i[:, written]) < TOL


def test_estimate_frame_families_agree():
    """row-lane estimate_frame == generic estimate_frame (same device functions, different FFT) on a big ragged batch."""
    import gfdm_amd
    rng = np.random.default_rng(9)
    for (M, K, A, dc_free) in ((9, 64, 52, True), (5, 32, 24, False), (31, 256, 220, True)):
        pre = (rng.standard_normal(2 * K) + 1j * rng.standard_normal(2 * K)) / np.sqrt(2)
        rx = rng.standard_normal((1031, 2 * K)) + 1j * rng.standard_normal((1031, 2 * K))
        fast = gfdm_amd.ChannelEstimator(M, K, A, dc_free, 1, pre)
        with gfdm_amd.generic_family_for_testing():
            slow = gfdm_amd.ChannelEstimator(M, K, A, dc_free, 1, pre)
        assert (fast.kernel_name(), slow.kernel_name()) == ("rowlane", "generic_lds")
        a, b = fast.estimate_frame(rx), slow.estimate_frame(rx)
        assert rel_err(a, b) < 1e-5
        assert rel_err(a, R.estimate_frame(rx, pre, M, K, A, dc_free)) < 1e-5


def test_estimator_feeds_zero_forcing_receiver_on_device():
    """preamble -> estimate_frame -> f_eq of the IC receiver, all device resident: transmitted symbols recovered through a
    frequency-selective channel (the chain of examples/hier_gfdm_receiver.grc)."""
    import torch
    import gfdm_amd
    from gfdm_amd import synth
    g = load_est_golden("est_cfg2_m9_k64_a52")
    M, K, A, L = g["M"], g["K"], g["A"], 2
    N = M * K
    dev = torch.device("cuda:0")
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    smap = g["smap"]
    B = 300
    h = g["channel"]
    tx = gfdm_amd.Transmitter(M, K, A, 0, 0, 0, smap, True, L, taps, np.zeros(0, complex), [0], [np.zeros(0, complex)])
    sym = synth.qpsk_symbols(11, B, A * M, dev)
    blocks = tx.modulate(sym)
    fh = torch.tensor(np.fft.fft(h, N), dtype=torch.complex64, device=dev)
    blocks_ch = torch.fft.ifft(torch.fft.fft(blocks, dim=-1) * fh, dim=-1).contiguous()
    rx_pre = torch.tensor(np.tile(g["rx_preambles"][1], (B, 1)), dtype=torch.complex64, device=dev)
    est = gfdm_amd.ChannelEstimator(M, K, A, True, 1, g["preamble"])
    f_eq = est.estimate_frame(rx_pre)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 4, R.qpsk_points())
    adv.configure_frames(N, 0, smap, True)
    rec = adv.demodulate_frames(blocks_ch, f_eq)
    torch.cuda.synchronize()
    assert f_eq.shape == (B, N) and rec.shape == sym.shape
    assert float((rec - sym).abs().max()) < 0.25
    assert np.array_equal(est.estimate_frame(rx_pre.cpu().numpy()), f_eq.cpu().numpy())      # host path == device path


def test_estimate_snr_against_oracle_and_known_level():
    import torch
    import gfdm_amd
    g = load_est_golden("est_cfg2_m9_k64_a52")
    K, A = g["K"], g["A"]
    rng = np.random.default_rng(3)
    clean = g["rx_preambles"][1]
    B = 256
    sigma = np.sqrt(np.mean(np.abs(clean) ** 2) / 100.0 / 2)
    rx = clean[None, :] + sigma * (rng.standard_normal((B, 2 * K)) + 1j * rng.standard_normal((B, 2 * K)))
    for dc_free in (True, False):
        est = gfdm_amd.ChannelEstimator(g["M"], K, A, dc_free, 1, g["preamble"] + (0 if dc_free else 0.01))
        snr, cnrs = est.estimate_snr(rx)
        ref_snr, ref_cnrs = R.estimate_snr(rx.astype(np.complex64), K, A, dc_free)
        assert snr.shape == (B,) and cnrs.shape == (B, A)
        assert np.max(np.abs(snr / ref_snr - 1.0)) < 1e-4
        assert np.max(np.abs(cnrs - ref_cnrs) / np.max(ref_cnrs)) < 1e-4
        s1, c1 = est.estimate_snr(rx[5])
        assert isinstance(s1, float) and s1 == snr[5] and np.array_equal(c1, cnrs[5])
        d_snr, d_cnrs = est.estimate_snr(torch.tensor(rx, dtype=torch.complex64, device="cuda:0"))
        torch.cuda.synchronize()
        assert np.array_equal(d_snr.cpu().numpy(), snr) and np.array_equal(d_cnrs.cpu().numpy(), cnrs)
    snr, _ = gfdm_amd.ChannelEstimator(g["M"], K, A, True, 1, g["preamble"]).estimate_snr(rx)
    assert abs(np.mean(snr) / (2 * 100.0 * K / A) - 1.0) < 0.1                   # 20 dB over the band, see test_oracle.py


@pytest.mark.parametrize("name", __import__("conftest").snr_golden_names())
def test_estimate_snr_reference_known_answer(name):
    """The reference's own estimate_snr test (python/qa_python_bindings.py:492-529: a 1024 / 936-subcarrier preamble in noise at 4 dB,
    estimate within 1 dB) and pygfdm.simulation.estimate_snr0 on the same vectors (tests/golden/make_golden_snr.py), through ctypes
    and the pybind11 class."""
    import gfdm_amd
    import gfdm_python
    from conftest import load_snr_golden
    g = load_snr_golden(name)
    est = gfdm_amd.ChannelEstimator(g["M"], g["K"], g["A"], True, 1, g["preamble"])
    snr, cnrs = est.estimate_snr(g["rx_preambles"])
    assert np.max(np.abs(snr / g["pygfdm_estimate_snr0"] - 1.0)) < 1e-4
    limit = 1.0 if g["K"] >= 1024 else 2.0
    assert np.max(np.abs(10 * np.log10(snr) - g["snr_db"])) < limit
    assert np.allclose(cnrs.sum(axis=-1), g["A"] * snr, rtol=1e-4)
    pest = gfdm_python.Preamble_channel_estimator(g["M"], g["K"], g["A"], True, 1, g["preamble"])
    res = pest.estimate_snr(g["rx_preambles"][0])                 # the call of the reference test
    assert abs(10 * np.log10(res) - g["snr_db"]) < limit and abs(res / g["pygfdm_estimate_snr0"][0] - 1.0) < 1e-4


def test_estimator_argument_errors():
    import gfdm_amd
    pre = np.ones(128, complex)
    with pytest.raises(ValueError, match="2 \\* fft_len"):
        gfdm_amd.ChannelEstimator(9, 64, 52, True, 1, pre[:100])
    with pytest.raises(ValueError, match="active_subcarriers"):
        gfdm_amd.ChannelEstimator(9, 64, 64, True, 1, pre)                       # dc-free needs a free bin
    with pytest.raises(ValueError, match="active_subcarriers"):
        gfdm_amd.ChannelEstimator(9, 64, 51, True, 1, pre)
    est = gfdm_amd.ChannelEstimator(9, 64, 52, True, 1, pre)
    with pytest.raises(RuntimeError, match="multiple"):
        est.estimate_frame(np.zeros(100, complex))


def test_estimator_pybind_surface():
    """gfdm_python.Preamble_channel_estimator (python/bindings/preamble_channel_estimator_python.cc:30-99) over the C++ class."""
    import gfdm_python
    g = load_est_golden("est_ref_m5_k64_a52")
    M, K, A = g["M"], g["K"], g["A"]
    est = gfdm_python.Preamble_channel_estimator(M, K, A, True, 1, g["preamble"])
    assert (est.timeslots(), est.subcarriers(), est.active_subcarriers(), est.frame_len(), est.is_dc_free()) == (M, K, A, M * K, True)
    one = est.estimate_frame(g["rx_preambles"][1])
    assert one.shape == (M * K,) and one.dtype == np.complex64
    assert_places(one, g["pygfdm_frame_estimates"][1], 4)
    batch = est.estimate_frame(g["rx_preambles"])
    assert batch.shape == (4, M * K) and np.array_equal(batch[1], one)
    snr = est.estimate_snr(g["rx_preambles"][3])
    ref_snr, ref_cnrs = R.estimate_snr(g["rx_preambles"][3].astype(np.complex64), K, A, True)
    assert isinstance(snr, float) and abs(snr / ref_snr - 1.0) < 1e-4
    snrs, cnrs = est.estimate_snr_cnrs(g["rx_preambles"])
    assert snrs.shape == (4,) and cnrs.shape == (4, A) and snrs[3] == np.float32(snr)
    assert np.max(np.abs(cnrs[3] - ref_cnrs)) / np.max(ref_cnrs) < 1e-4
    assert np.allclose(est.preamble_filter_taps(), R.gaussian_taps(), atol=1e-7)
    with pytest.raises(RuntimeError, match="MUST be equal to 2 \\* subcarriers"):
        est.estimate_frame(np.zeros(100, np.complex64))
    with pytest.raises(ValueError, match="2 \\* fft_len"):
        gfdm_python.Preamble_channel_estimator(M, K, A, True, 1, g["preamble"][:100])


@pytest.mark.parametrize("M,K,L,A,dc_free", [(9, 64, 2, 52, True), (5, 64, 2, 52, True), (15, 128, 2, 110, False), (5, 32, 2, 24, True), (9, 32, 2, 28, False), (15, 128, 4, 110, True),
                                             (31, 256, 2, 220, True), (7, 12, 2, 8, True), (3, 48, 2, 40, False), (127, 16, 2, 12, True)])
def test_receivers_with_fused_estimator(M, K, L, A, dc_free):
    """demodulate_estimated == estimate_frame followed by demodulate_equalize (the oracle's chain), for the plain receiver and the
    IC receiver, on plain blocks and on bursts (preamble + frame in one buffer, demapped output); row-lane and generic shapes (M = 127: the generic
    family with its transforms and cancellation rounds on the matrix cores, operands in the free LDS tile)."""
    import gfdm_amd
    rng = np.random.default_rng(M * K + A)
    N, B = M * K, 7
    taps = get_frequency_domain_filter("rrc", 0.3, M, K, L)
    nt = R.normalize_taps(taps, M)
    smap = _active_bins(K, A, dc_free)
    # known preamble: flat spectrum on every bin (invertible everywhere), two identical halves
    pre = np.tile(np.fft.ifft(np.exp(2j * np.pi * rng.random(K))) * np.sqrt(K), 2)
    h = np.array([1, .4 - .2j, .15j, .05])
    d = np.zeros((B, K, M), complex)
    d[:, smap, :] = ((1 - 2 * rng.integers(0, 2, (B, A, M))) + 1j * (1 - 2 * rng.integers(0, 2, (B, A, M)))) / np.sqrt(2)
    gains = np.exp(0.3j * np.arange(B)) * (1 + 0.1 * np.arange(B))             # every burst sees its own channel
    blocks = np.fft.ifft(np.fft.fft(R.modulate(d.reshape(B, N), nt, M, K, L), axis=-1) * np.fft.fft(h, N), axis=-1) * gains[:, None]
    rx_pre = np.tile(np.fft.ifft(np.fft.fft(pre[:K]) * np.fft.fft(h, K)), 2)[None, :] * gains[:, None]
    rx_pre = rx_pre + 1e-3 * (rng.standard_normal((B, 2 * K)) + 1j * rng.standard_normal((B, 2 * K)))
    feq = R.estimate_frame(rx_pre.astype(np.complex64), pre.astype(np.complex64), M, K, A, dc_free)
    est = gfdm_amd.ChannelEstimator(M, K, A, dc_free, 1, pre)
    dem = gfdm_amd.Demodulator(M, K, L, taps)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 2, R.qpsk_points())
    with pytest.raises(ValueError, match="set_channel_estimator"):
        dem.demodulate_estimated(blocks, rx_pre)
    ref_dem = R.demodulate(blocks, nt, M, K, L, feq)
    ref_adv = R.advanced_receive(blocks, nt, M, K, L, smap, R.qpsk_points(), 2, f_eq=feq, kind="qpsk")
    for rx, ref in ((dem, ref_dem), (adv, ref_adv)):
        rx.set_channel_estimator(est)
        got = rx.demodulate_estimated(blocks, rx_pre)
        assert got.shape == (B, N)
        check_err("fused_est_%d_%d_%d" % (M, K, A), rel_err(got, ref), TOL)
        # the two-kernel chain on the GPU gives the same symbols
        two_step = rx.demodulate_equalize(blocks.astype(np.complex64), est.estimate_frame(rx_pre))
        check_err("fused_est_vs_chain_%d_%d_%d" % (M, K, A), rel_err(got, two_step.reshape(B, N)), TOL)
    assert np.max(np.abs(adv.demodulate_estimated(blocks, rx_pre).reshape(B, K, M)[:, smap, :] - d[:, smap, :])) < 0.45    # decisions all correct (coarse estimate at small K)
    # bursts: [junk | preamble cp | core preamble | cp | block | cs], one buffer for both pointers
    pcp, cp, cs = K // 4, K // 8 + 1, 3
    blen = 5 + pcp + 2 * K + cp + N + cs
    bursts = rng.standard_normal((B, blen)) + 1j * rng.standard_normal((B, blen))
    bursts[:, 5 + pcp:5 + pcp + 2 * K] = rx_pre
    off = 5 + pcp + 2 * K + cp
    bursts[:, off:off + N] = blocks
    bursts = bursts.astype(np.complex64)
    for rx, ref in ((dem, ref_dem), (adv, ref_adv)):
        rx.configure_frames(blen, off, smap[::-1], True)
        got = rx.demodulate_estimated(bursts, bursts.ravel()[5 + pcp:], preamble_stride=blen)
        assert got.shape == (B, A * M)
        check_err("fused_est_burst_%d_%d_%d" % (M, K, A), rel_err(got, R.demap_from_resources(ref, M, K, smap, True)), TOL)
    est2 = gfdm_amd.ChannelEstimator(M + 1, K, A, dc_free, 1, pre)
    with pytest.raises(ValueError, match="estimator is for"):
        dem.set_channel_estimator(est2)


def test_fused_estimator_device_path_at_batch():
    import torch
    import gfdm_amd
    from gfdm_amd import synth
    g = load_est_golden("est_cfg2_m9_k64_a52")
    M, K, A, L = g["M"], g["K"], g["A"], 2
    N, B = M * K, 4099
    dev = torch.device("cuda:0")
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    smap = g["smap"]
    tx = gfdm_amd.Transmitter(M, K, A, 0, 0, 0, smap, True, L, taps, np.zeros(0, complex), [0], [np.zeros(0, complex)])
    sym = synth.qpsk_symbols(5, B, A * M, dev)
    fh = torch.tensor(np.fft.fft(g["channel"], N), dtype=torch.complex64, device=dev)
    blocks = torch.fft.ifft(torch.fft.fft(tx.modulate(sym), dim=-1) * fh, dim=-1).contiguous()
    rx_pre = torch.tensor(np.tile(g["rx_preambles"][2], (B, 1)), dtype=torch.complex64, device=dev)
    est = gfdm_amd.ChannelEstimator(M, K, A, True, 1, g["preamble"])
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 2, R.qpsk_points())
    assert adv.kernel_name() == "rowlane"
    adv.configure_frames(N, 0, smap, True)
    adv.set_channel_estimator(est)
    fused = adv.demodulate_estimated(blocks, rx_pre)
    chained = adv.demodulate_frames(blocks, est.estimate_frame(rx_pre))
    torch.cuda.synchronize()
    assert fused.shape == sym.shape
    assert float((fused - chained).abs().max()) < 2e-4
    assert float((fused - sym).abs().max()) < 0.3
    assert np.array_equal(adv.demodulate_estimated(blocks[:9].cpu().numpy(), rx_pre[:9].cpu().numpy()), fused[:9].cpu().numpy())
    adv.set_channel_estimator(None)
    with pytest.raises(ValueError, match="set_channel_estimator"):
        adv.demodulate_estimated(blocks, rx_pre)


def test_fused_estimator_pybind_surface():
    """gfdm_python.Demodulator / AdvancedReceiver .set_channel_estimator + .demodulate_estimated (C++ classes over the C-ABI)."""
    import gfdm_python
    g = load_est_golden("est_cfg2_m9_k64_a52")
    M, K, A, L = g["M"], g["K"], g["A"], 2
    N, B = M * K, 3
    rng = np.random.default_rng(1)
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L).astype(np.complex64)
    smap = g["smap"]
    blocks = (rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N))).astype(np.complex64)
    rx_pre = g["rx_preambles"][1:4].astype(np.complex64)
    feq = R.estimate_frame(rx_pre, g["preamble"].astype(np.complex64), M, K, A, True)
    est = gfdm_python.Preamble_channel_estimator(M, K, A, True, 1, g["preamble"])
    dem = gfdm_python.Demodulator(M, K, L, taps)
    dem.set_channel_estimator(est)
    got = dem.demodulate_estimated(blocks, rx_pre)
    assert got.shape == (B, N) and got.dtype == np.complex64
    check_err("fused_est_pybind_dem", rel_err(got, R.demodulate(blocks, R.normalize_taps(taps, M), M, K, L, feq)), TOL)
    adv = gfdm_python.AdvancedReceiver(M, K, L, taps, smap.tolist(), 2, gfdm_python.Constellation.qpsk(), 0)
    adv.configure_frames(N, 0, smap.tolist(), True, M)
    adv.set_channel_estimator(est)
    got = adv.demodulate_estimated(blocks, rx_pre)
    ref = R.advanced_receive(blocks, R.normalize_taps(taps, M), M, K, L, smap, R.qpsk_points(), 2, f_eq=feq, kind="qpsk")
    assert got.shape == (B, A * M)
    check_err("fused_est_pybind_adv", rel_err(got, R.demap_from_resources(ref, M, K, smap, True)), TOL)
    with pytest.raises(RuntimeError, match="rx_preamble size"):
        adv.demodulate_estimated(blocks, rx_pre[:2])
    adv.set_channel_estimator(None)
    with pytest.raises(ValueError, match="set_channel_estimator"):
        adv.demodulate_estimated(blocks, rx_pre)


def test_whole_chain_on_bursts_device_resident():
    """The complete chain of the reference's transmitter / receiver flowgraphs with TWO kernel launches, device resident:
    transmitter (mapper + modulator + prefix + preamble in front) -> multipath channel on the whole burst ->
    receiver (estimate from the burst's own preamble + prefix removal + ZF + IC + demapper).  Every symbol is recovered."""
    import torch
    import gfdm_amd
    from gfdm_amd import synth
    M, K, A, L = 9, 64, 52, 2
    N, B = M * K, 1000
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(21)
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    smap = _active_bins(K, A, True)
    spec = np.zeros(K, complex)
    spec[smap] = np.exp(2j * np.pi * rng.random(A))                     # known preamble: unit-modulus active bins, two identical halves
    core = np.tile(np.fft.ifft(spec) * np.sqrt(K), 2)
    pcp, cp, cs = 16, 16, 8
    preamble = np.concatenate((core[-pcp:], core))                      # its own cyclic prefix in front
    tx = gfdm_amd.Transmitter(M, K, A, cp, cs, 0, smap, True, L, taps, np.zeros(0, complex), [0], [preamble])
    blen = tx.output_vector_size()
    assert blen == pcp + 2 * K + cp + N + cs
    sym = synth.qpsk_symbols(77, B, A * M, dev)
    bursts = tx.transmit(sym)[0]
    h = np.array([1, .45 - .2j, .2j, .08, -.05j])                       # 5 taps < both prefixes
    hf = torch.tensor(np.fft.fft(h, 2 * blen), dtype=torch.complex64, device=dev)
    rx = torch.fft.ifft(torch.fft.fft(bursts, n=2 * blen, dim=-1) * hf, dim=-1)[:, :blen].contiguous()      # linear convolution per burst
    rx = rx * torch.exp(1j * torch.linspace(0, 3, B, device=dev))[:, None]                                   # and a phase per burst
    est = gfdm_amd.ChannelEstimator(M, K, A, True, 1, core)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 4, R.qpsk_points())
    adv.configure_frames(blen, pcp + 2 * K + cp, smap, True)
    adv.set_channel_estimator(est)
    got = adv.demodulate_estimated(rx, rx.reshape(-1)[pcp:], preamble_stride=blen)
    torch.cuda.synchronize()
    assert got.shape == sym.shape
    err = (got - sym).abs()
    assert float(err.max()) < 0.35 and float(err.mean()) < 0.05
    assert bool(torch.all(torch.sign(got.real) == torch.sign(sym.real))) and bool(torch.all(torch.sign(got.imag) == torch.sign(sym.imag)))
