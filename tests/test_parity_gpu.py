"""GPU parity tests: the HIP path, called through the C-ABI (ctypes, host and device entry points) and through
the pybind11 module `gfdm_python`, against the CPU oracle and the committed pygfdm golden vectors.

Tolerance (BASELINE.json north_star): relative L2 error per block <= 1e-5 against the float64 oracle; where the
reference's own tests state decimal places (python/qa_python_bindings.py) those are asserted as well.
IC outputs are compared only on blocks whose oracle decision margin exceeds DECISION_GUARD: a hard decision taken
on a component closer to zero than fp32 noise may legitimately flip.
"""
import os

import numpy as np
import pytest

import c_oracle
import gfdm_ref as R
from conftest import assert_places, check_err, golden_names, have_gpu, ic_golden_names, load_golden, load_ic_golden, rel_err
from gfdm_amd.filters import get_frequency_domain_filter

pytestmark = pytest.mark.gpu

TOL = 1e-5
DECISION_GUARD = 1e-4

# (M, K, L, alpha): BASELINE.json configs 1-5 and the shapes of the reference's tests
SHAPES = [(5, 32, 2, 0.5), (9, 64, 2, 0.2), (15, 128, 4, 0.2), (31, 256, 2, 0.1),
          (16, 4, 2, 0.35), (21, 128, 2, 0.35), (8, 4, 2, 0.5), (127, 16, 4, 0.5), (127, 16, 2, 0.5), (9, 32, 2, 0.5),
          (25, 96, 2, 0.35), (7, 6, 2, 0.3), (3, 10, 6, 0.3), (4, 16, 3, 0.4),
          (5, 64, 2, 0.5), (15, 64, 2, 0.2), (9, 128, 2, 0.2), (15, 128, 2, 0.35)]     # further row-lane shapes (reference demos / tests)


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    if not have_gpu():
        pytest.fail("no MI355X visible: the HIP path cannot run (there is no CPU fallback to test instead)")


def qpsk(rng, shape):
    return ((1 - 2 * rng.integers(0, 2, shape)) + 1j * (1 - 2 * rng.integers(0, 2, shape))) / np.sqrt(2)


def guarded(ref_stages, smap, K, M):
    """blocks whose every decided component (all IC iterations) is at least DECISION_GUARD away from zero"""
    keep = None
    for d in [ref_stages["d0"]] + ref_stages["iters"][:-1]:
        v = d.reshape(-1, K, M)[:, smap, :]
        ok = (np.minimum(np.abs(v.real), np.abs(v.imag)).reshape(v.shape[0], -1).min(axis=1) > DECISION_GUARD)
        keep = ok if keep is None else (keep & ok)
    return keep


# ---------------------------------------------------------------- golden vectors, pybind11 surface

@pytest.mark.parametrize("name", golden_names())
def test_golden_modulator(name):
    import gfdm_python
    g = load_golden(name)
    mod = gfdm_python.Modulator(g["M"], g["K"], g["L"], g["taps"])
    for sym, ref in ((g["symbols"], g["pygfdm_modulate"]), (g["gauss_symbols"], g["pygfdm_modulate_gauss"])):
        for b in range(sym.shape[0]):
            res = mod.modulate(sym[b])                      # complex128 in, forcecast, as the reference tests do
            assert res.dtype == np.complex64 and res.shape == (g["M"] * g["K"],)
            assert rel_err(res, ref[b]) < TOL
        assert rel_err(mod.modulate_batch(sym), ref) < TOL
    assert_places(mod.modulate(g["symbols"][0]), g["pygfdm_modulate"][0], 5)


@pytest.mark.parametrize("name", [n for n in golden_names() if "_l4" not in n])
def test_golden_demodulator(name):
    import gfdm_python
    g = load_golden(name)
    dem = gfdm_python.Demodulator(g["M"], g["K"], g["L"], g["taps"])
    for frame, ref in ((g["pygfdm_modulate"], g["pygfdm_demodulate"]), (g["gauss_symbols"], g["pygfdm_demodulate_gauss"])):
        for b in range(frame.shape[0]):
            assert rel_err(dem.demodulate(frame[b]), ref[b]) < TOL
        assert rel_err(dem.demodulate_batch(frame), ref) < TOL
    assert_places(dem.demodulate(g["pygfdm_modulate"][0]), g["pygfdm_demodulate"][0], 5)
    # equaliser: qa_python_bindings.py:365-386 with a frequency-selective channel instead of a flat phase
    got = dem.demodulate_equalize(g["frame_through_channel"][0], g["f_eq"][0])
    assert rel_err(got, g["pygfdm_demodulate"][0]) < TOL
    assert rel_err(dem.demodulate_batch(g["frame_through_channel"], g["f_eq"]), g["pygfdm_demodulate"]) < TOL


@pytest.mark.parametrize("name", __import__("conftest").rx_overlap_golden_names())
def test_golden_demodulator_any_overlap(name):
    """The receiver at overlap 4, 6, 8 (BASELINE configs[3], the reference's M=127 K=16 overlap-4 shape, ...) against pygfdm's
    overlap-generic model gfdm_demodulate_fft_loop (tests/golden/make_golden_rx_overlap.py): pybind11 per block and batched, ctypes,
    and the frequency-domain output S = DFT_M of the model's symbols."""
    import gfdm_amd
    import gfdm_python
    from conftest import load_rx_overlap_golden
    g = load_rx_overlap_golden(name)
    M, K, L = g["M"], g["K"], g["L"]
    dem = gfdm_python.Demodulator(M, K, L, g["taps"])
    hd = gfdm_amd.Demodulator(M, K, L, g["taps"])
    for x, ref in ((g["frames"], g["pygfdm_demodulate_fft_loop"]), (g["gauss"], g["pygfdm_demodulate_fft_loop_gauss"])):
        for b in range(x.shape[0]):
            check_err("golden_rx_overlap_%s" % name, rel_err(dem.demodulate(x[b]), ref[b]), TOL)
        check_err("golden_rx_overlap_%s" % name, rel_err(dem.demodulate_batch(x), ref), TOL)
        check_err("golden_rx_overlap_%s" % name, rel_err(hd.demodulate(x), ref), TOL)
        S = np.fft.fft(ref.reshape(-1, K, M), axis=-1).reshape(ref.shape)
        check_err("golden_rx_overlap_S_%s" % name, rel_err(hd.fft_filter_downsample(x), S), TOL)
    assert_places(dem.demodulate(g["frames"][0]), g["pygfdm_demodulate_fft_loop"][0], 5)


@pytest.mark.parametrize("name", ic_golden_names())
def test_golden_ic_stage(name):
    """The IC stage against the reference's Python model (tests/golden/make_golden_ic.py; python/pygfdm/gfdm_receiver.py:99-114):
    ic_filter_taps, cancel_sc_interference (real RRC taps -> real-symmetric kernel; complex taps -> general kernel), and the
    receiver + IC rounds 1..5 with MF input (advanced_receiver_kernel_cc::generic_work) and ZF input (generic_work_equalize)."""
    import gfdm_amd
    import gfdm_python
    g = load_ic_golden(name)
    M, K, L = g["M"], g["K"], g["L"]
    N = M * K
    for taps, ic_ref, cancel_ref in ((g["taps"], g["pygfdm_ic_taps"], g["pygfdm_cancel"]),
                                     (g["ctaps"], g["pygfdm_ic_ctaps"], g["pygfdm_cancel_ctaps"])):
        dem = gfdm_python.Demodulator(M, K, L, taps)
        hd = gfdm_amd.Demodulator(M, K, L, taps)
        assert rel_err(np.array(hd.ic_filter_taps()), ic_ref) < 1e-6
        for b in range(g["td_in"].shape[0]):
            assert rel_err(dem.cancel_sc_interference(g["td_in"][b], g["fd_in"][b]), cancel_ref[b]) < TOL
        assert rel_err(hd.cancel_sc_interference(g["td_in"], g["fd_in"]), cancel_ref) < TOL
    hd = gfdm_amd.Demodulator(M, K, L, g["taps"])
    assert rel_err(hd.fft_filter_downsample(g["frames"]), g["S"]) < TOL
    assert rel_err(hd.transform_subcarriers_to_td(g["S"]), g["pygfdm_d0"]) < TOL
    B = g["frames"].shape[0]
    feq = np.fft.fft(np.array([1, .5, .1j, .1 + .05j]), N)[None, :] * np.exp(0.01j * np.arange(B))[:, None]
    frames_ch = np.fft.ifft(np.fft.fft(g["frames"], axis=-1) * feq, axis=-1)
    rounds = g["pygfdm_ic_iters"]
    qp = gfdm_python.Constellation.qpsk()
    for n in range(1, rounds.shape[0] + 1):
        for decision in ("auto", "nearest"):
            adv = gfdm_amd.AdvancedReceiver(M, K, L, g["taps"], np.arange(K), n, R.qpsk_points(), decision=decision)
            check_err("golden_ic_mf_%s_%s" % (name, decision), rel_err(adv.demodulate(g["frames"]), rounds[n - 1]), TOL)     # MF + IC
            # ZF + IC: the channel is applied in float64 and divided out again in float32 by the kernel
            check_err("golden_ic_zf_%s_%s" % (name, decision), rel_err(adv.demodulate_equalize(frames_ch, feq), rounds[n - 1]), TOL)
        padv = gfdm_python.AdvancedReceiver(M, K, L, g["taps"], list(range(K)), n, qp, 0)
        assert rel_err(padv.demodulate(g["frames"]), rounds[n - 1]) < TOL
    assert_places(padv.demodulate(g["frames"]), rounds[-1], 5)


# ---------------------------------------------------------------- the reference's own binding tests, restated

def test_ref_demodulator_init():
    """qa_python_bindings.py:304-319 (K = 96 is not a power of two)."""
    import gfdm_python
    M, K, L = 25, 96, 2
    taps = get_frequency_domain_filter("rrc", 0.35, M, K, L)
    demod = gfdm_python.Demodulator(M, K, L, taps)
    assert (demod.timeslots(), demod.subcarriers(), demod.overlap(), demod.block_size()) == (M, K, L, M * K)
    assert_places(np.array(demod.filter_taps()), taps, 6)


def test_ref_demodulator_flat_phase_equalize():
    """qa_python_bindings.py:365-386."""
    import gfdm_python
    g = load_golden("ref_m21_k128")
    dem = gfdm_python.Demodulator(g["M"], g["K"], g["L"], g["taps"])
    frame, ref = g["pygfdm_modulate"][0], g["pygfdm_demodulate"][0]
    res = dem.demodulate_equalize(frame * np.exp(1j), np.ones(ref.size, ref.dtype) * np.exp(1j))
    assert_places(res, ref, 5)


def test_ref_steps_and_genie_ic():
    """qa_python_bindings.py:388-440."""
    import gfdm_python
    g = load_golden("ref_m5_k32_a35")
    dem = gfdm_python.Demodulator(g["M"], g["K"], g["L"], g["taps"])
    data, frame, ref = g["symbols"][0], g["pygfdm_modulate"][0], g["pygfdm_demodulate"][0]
    fd_res = dem.fft_filter_downsample(frame)
    assert_places(dem.transform_subcarriers_to_td(fd_res), ref, 5)
    for _ in range(2):
        res = dem.transform_subcarriers_to_td(dem.cancel_sc_interference(data, fd_res))
    assert_places(res, data, 1)
    eq = np.ones(ref.size, ref.dtype) * np.exp(1j)
    fd_eq = dem.fft_equalize_filter_downsample(frame * np.exp(1j), eq)
    assert_places(dem.transform_subcarriers_to_td(fd_eq), ref, 5)


def test_ref_binding_error_messages():
    """python/bindings/modulator_python.cc:44-52, demodulator_python.cc:49-57,118-132."""
    import gfdm_python
    taps = get_frequency_domain_filter("rrc", 0.5, 5, 32, 2)
    mod, dem = gfdm_python.Modulator(5, 32, 2, taps), gfdm_python.Demodulator(5, 32, 2, taps)
    with pytest.raises(RuntimeError, match="Only ONE-dimensional vectors allowed!"):
        mod.modulate(np.zeros((2, 80), np.complex64))
    with pytest.raises(RuntimeError, match=r"Input vector size\(159\) MUST be equal to Modulator.block_size\(160\)!"):
        mod.modulate(np.zeros(159, np.complex64))
    with pytest.raises(RuntimeError, match=r"Input vector size\(161\) MUST be equal to Modulator.block_size\(160\)!"):
        dem.demodulate(np.zeros(161, np.complex64))
    with pytest.raises(RuntimeError, match=r"Channel vector size\(10\) MUST be equal to Demodulator.block_size\(160\)!"):
        dem.demodulate_equalize(np.zeros(160, np.complex64), np.ones(10, np.complex64))
    with pytest.raises(RuntimeError, match="Only ONE-dimensional vectors allowed!"):
        dem.cancel_sc_interference(np.zeros((160, 1), np.complex64), np.zeros(160, np.complex64))


def test_ref_advanced_receiver_loopbacks():
    """qa_advanced_receiver_sb_cc.py:45-82 (ic=0 equals plain receiver, 4 places), :84-119 (alpha=1, ic=64, 2 places),
    :121-132 (set_ic/get_ic), :134-172 (20 active subcarriers, data*2, ic=64: signs recovered)."""
    import gfdm_python
    qp = gfdm_python.Constellation.qpsk()
    g = load_golden("ref_m127_k16_l2")
    M, K, L = g["M"], g["K"], g["L"]
    adv = gfdm_python.AdvancedReceiver(M, K, L, g["taps"], list(range(K)), 0, qp, 0)
    assert_places(adv.demodulate_equalize(g["gauss_symbols"], np.ones_like(g["gauss_symbols"])), g["pygfdm_demodulate_gauss"], 4)

    g = load_golden("ref_m9_k64_a100")
    M, K, L = g["M"], g["K"], g["L"]
    mod = gfdm_python.Modulator(M, K, L, g["taps"])
    adv = gfdm_python.AdvancedReceiver(M, K, L, g["taps"], list(range(K)), 64, qp, 0)
    assert_places(adv.demodulate(mod.modulate_batch(g["symbols"])), g["symbols"], 2)

    g = load_golden("ref_m9_k32_act20")
    M, K, L = g["M"], g["K"], g["L"]
    adv = gfdm_python.AdvancedReceiver(M, K, L, g["taps"], g["smap"].tolist(), 64, qp, 0)
    adv.set_ic(2)
    assert adv.get_ic() == 2
    adv.set_ic(64)
    adv.set_phase_compensation(1)
    assert adv.get_phase_compensation() == 1
    adv.set_phase_compensation(0)
    mod = gfdm_python.Modulator(M, K, L, g["taps"])
    data = 2.0 * g["symbols"]
    res = adv.demodulate(mod.modulate_batch(data)).reshape(-1, K, M)[:, g["smap"], :]
    ref = data.reshape(-1, K, M)[:, g["smap"], :]
    assert np.all(np.sign(res.real) == np.sign(ref.real)) and np.all(np.sign(res.imag) == np.sign(ref.imag))


# ---------------------------------------------------------------- oracle parity, every entry point, every shape

@pytest.mark.parametrize("M,K,L,alpha", SHAPES)
def test_every_entry_point_against_oracle(M, K, L, alpha):
    import gfdm_amd
    rng = np.random.default_rng(1000 * M + K + L)
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    nt = R.normalize_taps(taps, M)
    N, B = M * K, 7
    mod, dem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
    assert rel_err(mod.filter_taps(), nt) < 1e-6 and rel_err(dem.filter_taps(), nt) < 1e-6
    assert rel_err(dem.ic_filter_taps(), R.ic_filter_taps(nt, M, L)) < 1e-6
    d = qpsk(rng, (B, N))
    x = R.modulate(d, nt, M, K, L)
    assert rel_err(mod.modulate(d), x) < TOL
    gauss = rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N))
    assert rel_err(mod.modulate(gauss), R.modulate(gauss, nt, M, K, L)) < TOL
    feq = np.fft.fft(np.array([1, .5, .1j, .1 + .05j]), N)[None, :] * np.exp(0.01j * np.arange(B))[:, None]
    xe = np.fft.ifft(np.fft.fft(x, axis=-1) * feq, axis=-1)
    S = R.fft_filter_downsample(x, nt, M, K, L)
    assert rel_err(dem.fft_filter_downsample(x), S) < TOL
    assert rel_err(dem.fft_equalize_filter_downsample(xe, feq), R.fft_filter_downsample(xe, nt, M, K, L, feq)) < TOL
    assert rel_err(dem.transform_subcarriers_to_td(S), R.transform_subcarriers_to_td(S, M, K)) < TOL
    assert rel_err(dem.cancel_sc_interference(d, S), R.cancel_sc_interference(d, S, R.ic_filter_taps(nt, M, L), M, K)) < TOL
    assert rel_err(dem.demodulate(x), R.demodulate(x, nt, M, K, L)) < TOL
    assert rel_err(dem.demodulate(gauss), R.demodulate(gauss, nt, M, K, L)) < TOL
    assert rel_err(dem.demodulate_equalize(xe, feq), R.demodulate(xe, nt, M, K, L, feq)) < TOL
    # and the plain-C float32 oracle agrees with the GPU to float32 noise as well
    co = c_oracle.COracle(M, K, L, taps)
    assert rel_err(dem.demodulate_equalize(xe, feq), co.demodulate(xe, feq)) < TOL


@pytest.mark.parametrize("M,K,family", [(9, 64, "generic_lds"), (7, 12, "generic_lds"), (8, 4, "generic_lds"), (21, 37, "generic_lds"), (5, 32, "generic_lds"),
                                        (127, 16, "generic_rader")])
def test_modulator_overlap_1_on_every_family_that_serves_it(M, K, family):
    """Overlap 1 (part_len = M / 2, lib/modulator_kernel_cc.cc:101,116-132) outside the Rader shape: the compiled and the run-time instantiated row-lane kernels
    start at overlap 2, so these shapes run on k_generic_modulate -- asserted, so that the family that served them is on record.  Expectation: the oracle AND the
    lines written out (tests/test_oracle.py overlap_1_by_definition); pygfdm cannot produce this case (see there).  The receivers refuse overlap 1 as the reference does."""
    import gfdm_amd
    from test_oracle import overlap_1_by_definition
    rng = np.random.default_rng(31 * M + K)
    taps = get_frequency_domain_filter("rrc", 0.3, M, K, 2)[:M] * np.exp(0.3j * np.arange(M))
    nt = R.normalize_taps(taps, M)
    mod = gfdm_amd.Modulator(M, K, 1, taps)
    assert mod.kernel_name() == family
    assert rel_err(mod.filter_taps(), nt) < 1e-6
    for B in (1, 6, 130):
        d = qpsk(rng, (B, M * K)) + 0.2 * (rng.standard_normal((B, M * K)) + 1j * rng.standard_normal((B, M * K)))
        got = mod.modulate(d)
        check_err("mod_L1_%d_%d_B%d" % (M, K, B), rel_err(got, R.modulate(d, nt, M, K, 1)), TOL)
        assert rel_err(got, overlap_1_by_definition(d, nt, M, K)) < TOL
    import gfdm_python
    assert rel_err(gfdm_python.Modulator(M, K, 1, taps).modulate(d[0]), R.modulate(d[0], nt, M, K, 1)) < TOL
    for make in (lambda: gfdm_amd.Demodulator(M, K, 1, taps), lambda: gfdm_amd.AdvancedReceiver(M, K, 1, taps, np.arange(K), 1, R.qpsk_points()),
                 lambda: gfdm_python.Demodulator(M, K, 1, taps)):
        with pytest.raises(ValueError, match="overlap MUST be greater or equal 2"):          # lib/receiver_kernel_cc.cc:48-52
            make()
    assert gfdm_amd.capi.EINVAL_OVERLAP == -2                                               # GFDM_HIP_EINVAL_OVERLAP, include/gfdm_hip.h


@pytest.mark.parametrize("M,K,L,alpha", SHAPES)
@pytest.mark.parametrize("pc", [0, 1])
def test_advanced_receiver_against_oracle(M, K, L, alpha, pc):
    import gfdm_amd
    rng = np.random.default_rng(77 * M + K + L + pc)
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    nt = R.normalize_taps(taps, M)
    N, B = M * K, 6
    smap = np.arange(K) if K < 8 else np.concatenate((np.arange(1, K // 2 - 1), np.arange(K // 2 + 2, K)))
    d = np.zeros((B, K, M), complex)
    d[:, smap, :] = qpsk(rng, (B, len(smap), M))
    x = R.modulate(d.reshape(B, N), nt, M, K, L)
    feq = np.fft.fft(np.array([1, .5, .1j, .1 + .05j]), N)[None, :] * np.exp(0.01j * np.arange(B))[:, None]
    xe = np.fft.ifft(np.fft.fft(x, axis=-1) * feq, axis=-1)
    checked = checked_mf = 0
    for ic_iter in (0, 1, 2, 5):
        for kind, pts in (("qpsk", R.qpsk_points()), ("nearest", R.qpsk_points() * np.exp(0.1j))):
            adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, ic_iter, pts, do_phase_compensation=pc,
                                            decision="auto" if kind == "qpsk" else "nearest")
            ref, st = R.advanced_receive(xe, nt, M, K, L, smap, pts, ic_iter, f_eq=feq, do_phase_compensation=pc, kind=kind,
                                         return_stages=True)
            got = adv.demodulate_equalize(xe, feq)
            keep = guarded(st, smap, K, M) if ic_iter > 0 else np.ones(B, bool)
            checked += int(keep.sum())
            check_err("adv_zf_pc%d_%d_%d_%d_%s" % (pc, M, K, L, kind), rel_err(got[keep], ref[keep]), TOL)
            # the unequalised (MF) input through the same receiver: its own stages, its own decision guard
            ref0, st0 = R.advanced_receive(x, nt, M, K, L, smap, pts, ic_iter, do_phase_compensation=pc, kind=kind, return_stages=True)
            got0 = adv.demodulate(x)
            keep0 = guarded(st0, smap, K, M) if ic_iter > 0 else np.ones(B, bool)
            checked_mf += int(keep0.sum())
            check_err("adv_mf_pc%d_%d_%d_%d_%s" % (pc, M, K, L, kind), rel_err(got0[keep0], ref0[keep0]), TOL)
    assert checked >= 6 * B and checked_mf >= 6 * B        # the guard may drop a few blocks, never most of them


# ---------------------------------------------------------------- device-pointer (batched, asynchronous) entry points

@pytest.mark.parametrize("M,K,L,alpha", SHAPES[:4])
def test_device_entry_points_match_host_entry_points(M, K, L, alpha):
    import torch
    import gfdm_amd
    from gfdm_amd import synth
    dev = torch.device("cuda:0")
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    N, B = M * K, 33
    mod, dem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points())
    sym = synth.qpsk_symbols(5, B, N, dev)
    feq = synth.channel_response(5, B, N, dev)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        x = mod.modulate(sym)
        xe = synth.through_channel(x, feq)
        outs = [dem.demodulate(x), dem.demodulate_equalize(xe, feq), dem.fft_filter_downsample(x),
                dem.fft_equalize_filter_downsample(xe, feq), adv.demodulate(x), adv.demodulate_equalize(xe, feq)]
        S = outs[2]
        outs += [dem.transform_subcarriers_to_td(S), dem.cancel_sc_interference(sym, S)]
    stream.synchronize()
    sym_h, x_h, xe_h, feq_h, S_h = (t.cpu().numpy() for t in (sym, x, xe, feq, S))
    host = [dem.demodulate(x_h), dem.demodulate_equalize(xe_h, feq_h), dem.fft_filter_downsample(x_h),
            dem.fft_equalize_filter_downsample(xe_h, feq_h), adv.demodulate(x_h), adv.demodulate_equalize(xe_h, feq_h),
            dem.transform_subcarriers_to_td(S_h), dem.cancel_sc_interference(sym_h, S_h)]
    assert np.array_equal(mod.modulate(sym_h), x_h)
    for d_out, h_out in zip(outs, host):
        assert np.array_equal(d_out.cpu().numpy(), h_out)       # same kernels, same inputs: bit-identical
    nt = R.normalize_taps(taps, M)
    assert rel_err(x_h, R.modulate(sym_h, nt, M, K, L)) < TOL


def test_empty_and_ragged_batches():
    import gfdm_amd
    taps = get_frequency_domain_filter("rrc", 0.2, 9, 64, 2)
    mod, dem = gfdm_amd.Modulator(9, 64, 2, taps), gfdm_amd.Demodulator(9, 64, 2, taps)
    assert mod.modulate(np.zeros((0, 576), np.complex64)).shape == (0, 576)
    assert dem.demodulate(np.zeros(0, np.complex64)).shape == (0,)
    with pytest.raises(RuntimeError, match="multiple of block_size"):
        mod.modulate(np.zeros(577, np.complex64))
    rng = np.random.default_rng(3)
    nt = R.normalize_taps(taps, 9)
    for B in (1, 2, 3, 5, 63, 64, 65, 257):                    # batch sizes that do not fill a workgroup / wave evenly
        d = qpsk(rng, (B, 576))
        x = mod.modulate(d)
        assert rel_err(x, R.modulate(d, nt, 9, 64, 2)) < TOL
        assert rel_err(dem.demodulate(x), R.demodulate(x, nt, 9, 64, 2)) < TOL


# ---------------------------------------------------------------- full BASELINE sizes: size-independent properties

FULL = [("cfg2", 9, 64, 2, 0.2, 4096), ("cfg4", 15, 128, 4, 0.2, 8192), ("cfg5", 31, 256, 2, 0.1, 8192)]     # cfg4 / cfg5: the per-GPU share of 65 536 blocks


@pytest.mark.parametrize("name,M,K,L,alpha,B", FULL)
def test_full_size_properties(name, M, K, L, alpha, B):
    """At BASELINE batch sizes the float64 oracle is too slow to run everywhere, so check (1) random blocks against
    the oracle, (2) linearity of modulator and receiver, (3) the transpose identity between them, (4) IC loop-back
    recovers the transmitted QPSK symbols, (5) equaliser round trip."""
    import torch
    import gfdm_amd
    from gfdm_amd import synth
    dev = torch.device("cuda:0")
    N = M * K
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    nt = R.normalize_taps(taps, M)
    mod, dem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points())
    sym = synth.qpsk_symbols(0, B, N, dev)
    feq = synth.channel_response(0, B, N, dev)
    x = mod.modulate(sym)
    y = dem.demodulate(x)
    xe = synth.through_channel(x, feq)
    ye = dem.demodulate_equalize(xe, feq)
    z = adv.demodulate_equalize(xe, feq)
    torch.cuda.synchronize()
    pick = np.unique(np.concatenate(([0, 1, B - 1], np.random.default_rng(5).integers(0, B, 13))))
    sym_h = sym[pick].cpu().numpy()
    assert rel_err(x[pick].cpu().numpy(), R.modulate(sym_h, nt, M, K, L)) < TOL
    assert rel_err(y[pick].cpu().numpy(), R.demodulate(x[pick].cpu().numpy(), nt, M, K, L)) < TOL
    ref_ic = R.advanced_receive(xe[pick].cpu().numpy(), nt, M, K, L, np.arange(K), R.qpsk_points(), 2, f_eq=feq[pick].cpu().numpy(), kind="qpsk")
    assert rel_err(z[pick].cpu().numpy(), ref_ic) < TOL
    # MF + IC (BASELINE configs[3]'s stated mode): the unequalised frames through the IC receiver, strict against the oracle
    zmf = adv.demodulate(x)
    torch.cuda.synchronize()
    ref_mf, st_mf = R.advanced_receive(x[pick].cpu().numpy(), nt, M, K, L, np.arange(K), R.qpsk_points(), 2, kind="qpsk", return_stages=True)
    keep = guarded(st_mf, np.arange(K), K, M)
    assert keep.sum() >= len(pick) - 1
    assert rel_err(zmf[pick].cpu().numpy()[keep], ref_mf[keep]) < TOL
    assert bool(torch.all(torch.sign(zmf.real) == torch.sign(sym.real))) and bool(torch.all(torch.sign(zmf.imag) == torch.sign(sym.imag)))
    # equaliser round trip over the whole batch (fp32 channel application + division: 1e-4 is its own noise floor)
    assert float((ye - y).abs().max() / y.abs().max()) < 1e-4
    # IC loop-back: every symbol of every block lands on the transmitted constellation point's quadrant and close to it
    assert bool(torch.all(torch.sign(z.real) == torch.sign(sym.real))) and bool(torch.all(torch.sign(z.imag) == torch.sign(sym.imag)))
    assert float((z - sym).abs().max()) < 0.2
    # linearity: demod(a*x1 + b*x2) == a*demod(x1) + b*demod(x2)
    half = B // 2
    a, b = 0.75 - 0.5j, -1.25 + 0.25j
    mix = (a * x[:half] + b * x[half:2 * half]).contiguous()
    lin = dem.demodulate(mix)
    assert float((lin - (a * y[:half] + b * y[half:2 * half])).abs().max() / y.abs().max()) < 2e-5
    # transpose identity on the device results: <conj(R x), d> == (N/M) <conj(x), A_conj d>
    modc = gfdm_amd.Modulator(M, K, L, np.conj(taps))
    g = torch.randn(B, N, dtype=torch.complex64, device=dev)
    w = torch.randn(B, N, dtype=torch.complex64, device=dev)
    lhs = (dem.demodulate(g).conj().to(torch.complex128) * w).sum(dim=-1)
    rhs = (N / M) * (g.conj().to(torch.complex128) * modc.modulate(w)).sum(dim=-1)
    assert float(((lhs - rhs).abs() / lhs.abs().clamp_min(1e-3)).max()) < 1e-3


def test_baseline_shapes_run_on_the_tuned_family():
    """BASELINE configs 1-5 must be served by the tuned row-lane kernels, every other shape by the generic family
    (both are HIP; this guards against silently benchmarking the slow path)."""
    import gfdm_amd
    for (M, K, L, alpha) in SHAPES[:4]:
        taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
        assert gfdm_amd.Modulator(M, K, L, taps).kernel_name() == "rowlane"
        assert gfdm_amd.Demodulator(M, K, L, taps).kernel_name() == "rowlane"
        assert gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points()).kernel_name() == "rowlane"
    taps = get_frequency_domain_filter("rrc", 0.35, 127, 16, 2)                  # M > 48: no register codelet; the reference's QA shape: Rader transforms
    assert gfdm_amd.Demodulator(127, 16, 2, taps).kernel_name() == "generic_rader"
    taps = get_frequency_domain_filter("rrc", 0.35, 113, 16, 2)
    assert gfdm_amd.Demodulator(113, 16, 2, taps).kernel_name() == "generic_lds"
    taps = get_frequency_domain_filter("rrc", 0.35, 9, 74, 2)                    # K = 2 x 37: a prime factor above 32, no butterfly codelet
    assert gfdm_amd.Demodulator(9, 74, 2, taps).kernel_name() == "generic_lds"


@pytest.mark.parametrize("M,K,L,alpha", SHAPES[:4])
def test_generic_family_on_the_tuned_shapes(M, K, L, alpha):
    """The generic kernel family forced (test hook) onto BASELINE shapes 1-5: every mode against the oracle and against the row-lane
    family, so both HIP families are checked on the same inputs."""
    import gfdm_amd
    rng = np.random.default_rng(M + K)
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    nt = R.normalize_taps(taps, M)
    N, B = M * K, 9
    allk = np.arange(K)
    with gfdm_amd.generic_family_for_testing():
        gmod, gdem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
        gadv = gfdm_amd.AdvancedReceiver(M, K, L, taps, allk, 2, R.qpsk_points())
    rmod, rdem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
    radv = gfdm_amd.AdvancedReceiver(M, K, L, taps, allk, 2, R.qpsk_points())
    assert (gmod.kernel_name(), gdem.kernel_name(), gadv.kernel_name()) == ("generic_lds",) * 3
    assert (rmod.kernel_name(), rdem.kernel_name(), radv.kernel_name()) == ("rowlane",) * 3
    d = qpsk(rng, (B, N))
    x = R.modulate(d, nt, M, K, L)
    feq = np.fft.fft(np.array([1, .5, .1j, .1 + .05j]), N)[None, :] * np.exp(0.01j * np.arange(B))[:, None]
    xe = np.fft.ifft(np.fft.fft(x, axis=-1) * feq, axis=-1)
    assert rel_err(gmod.modulate(d), x) < TOL and rel_err(gmod.modulate(d), rmod.modulate(d)) < TOL
    assert rel_err(gdem.demodulate(x), R.demodulate(x, nt, M, K, L)) < TOL
    assert rel_err(gdem.demodulate_equalize(xe, feq), rdem.demodulate_equalize(xe, feq)) < TOL
    assert rel_err(gdem.fft_equalize_filter_downsample(xe, feq), R.fft_filter_downsample(xe, nt, M, K, L, feq)) < TOL
    for inp, eq in ((x, None), (xe, feq)):
        ref, st = R.advanced_receive(inp, nt, M, K, L, allk, R.qpsk_points(), 2, f_eq=eq, kind="qpsk", return_stages=True)
        keep = guarded(st, allk, K, M)
        assert keep.sum() >= B - 1
        g = gadv.demodulate(inp) if eq is None else gadv.demodulate_equalize(inp, eq)
        r = radv.demodulate(inp) if eq is None else radv.demodulate_equalize(inp, eq)
        assert rel_err(g[keep], ref[keep]) < TOL and rel_err(r[keep], ref[keep]) < TOL


def test_rader_timeslot_transforms_match_the_dense_forms_and_the_oracle():
    """The reference's QA shape (python/qa_simple_receiver_cc.py:58-83, qa_simple_modulator_cc.py: 127 timeslots, 16 subcarriers) runs its plain
    blocks on the Rader kernels (csrc/gfdm_rader.hip): every entry point they serve against the float64 oracle at 1e-5 and against the dense
    transforms of the generic kernels -- vector ALU (mode 0) and matrix cores (mode 2) -- at 2e-6, for overlap 2 and 4 (filter in registers), 3 and 6
    (run-time filter loop), real and complex taps, batches that do not fill a wave generation; the advanced receiver's cancellation rounds on the same
    kernels; the paths the Rader kernels do not serve (frames + demapper) still answer correctly on such a handle."""
    import gfdm_amd
    M, K = 127, 16
    N = M * K
    rng = np.random.default_rng(127)
    for L, alpha, cplx in ((2, 0.5, False), (4, 0.5, False), (3, 0.3, False), (6, 0.2, True)):
        taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
        if cplx:
            taps = taps * np.exp(2j * np.pi * rng.random(M * L)) * (1 + 0.2 * rng.standard_normal(M * L))
        nt = R.normalize_taps(taps, M)
        fam = {}
        for mode, want in ((1, "generic_rader"), (0, "generic_lds"), (2, "generic_lds")):
            prev = gfdm_amd.set_dft_matrix_cores(mode)
            try:
                fam[mode] = (gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps))
            finally:
                gfdm_amd.set_dft_matrix_cores(prev)
            assert fam[mode][0].kernel_name() == want and fam[mode][1].kernel_name() == want
        for B in (1, 5, 300):
            d = qpsk(rng, (B, N)) + 0.1 * (rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N)))
            x = R.modulate(d, nt, M, K, L) + 0.05 * (rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N)))
            feq = np.fft.fft(np.array([1, .5, .1j, .1 + .05j]), N)[None, :] * np.exp(0.01j * np.arange(B))[:, None]
            ref = {"mod": R.modulate(d, nt, M, K, L), "mf": R.demodulate(x, nt, M, K, L), "zf": R.demodulate(x, nt, M, K, L, f_eq=feq),
                   "fd": R.fft_filter_downsample(x, nt, M, K, L), "fdeq": R.fft_filter_downsample(x, nt, M, K, L, feq)}
            got = {}
            for mode, (mod, dem) in fam.items():
                got[mode] = {"mod": mod.modulate(d), "mf": dem.demodulate(x), "zf": dem.demodulate_equalize(x, feq),
                             "fd": dem.fft_filter_downsample(x), "fdeq": dem.fft_equalize_filter_downsample(x, feq)}
            for k in ref:
                check_err("rader_%s_L%d_B%d" % (k, L, B), rel_err(got[1][k], ref[k]), TOL)
                assert rel_err(got[1][k], got[0][k]) < 2e-6 and rel_err(got[1][k], got[2][k]) < 2e-6
    # overlap 1 (modulator only: the receiver needs overlap >= 2): half of the bins of a tap part are used, lib/modulator_kernel_cc.cc:101
    taps1 = get_frequency_domain_filter("rrc", 0.5, M, K, 2)[:M] * np.exp(0.4j * np.arange(M))
    d = qpsk(rng, (7, N))
    mod1 = gfdm_amd.Modulator(M, K, 1, taps1)
    assert mod1.kernel_name() == "generic_rader"
    check_err("rader_mod_L1", rel_err(mod1.modulate(d), R.modulate(d, R.normalize_taps(taps1, M), M, K, 1)), TOL)
    # the advanced receiver's cancellation rounds (two more row transforms per round, S in a third tile): MF and ZF input, a partial subcarrier map, 1-3 rounds,
    # both decision rules, with and without phase compensation -- against the oracle and against the dense form of the generic kernels
    taps = get_frequency_domain_filter("rrc", 0.3, M, K, 2)
    nt = R.normalize_taps(taps, M)
    smap = np.arange(1, K - 1)
    B = 5
    dsym = np.zeros((B, K, M), complex)
    dsym[:, smap, :] = qpsk(rng, (B, len(smap), M))
    x = R.modulate(dsym.reshape(B, N), nt, M, K, 2) * np.exp(0.04j)
    feq = np.fft.fft(np.array([1, .5, .1j, .1 + .05j]), N)[None, :] * np.exp(0.01j * np.arange(B))[:, None]
    xe = np.fft.ifft(np.fft.fft(x, axis=-1) * feq, axis=-1)
    for rounds, dec, pc in ((1, "auto", 0), (2, "auto", 0), (3, "nearest", 0), (2, "auto", 1), (2, "nearest", 1)):
        adv = gfdm_amd.AdvancedReceiver(M, K, 2, taps, smap, rounds, R.qpsk_points(), do_phase_compensation=pc, decision=dec)
        prev = gfdm_amd.set_dft_matrix_cores(2)
        try:
            dense = gfdm_amd.AdvancedReceiver(M, K, 2, taps, smap, rounds, R.qpsk_points(), do_phase_compensation=pc, decision=dec)
        finally:
            gfdm_amd.set_dft_matrix_cores(prev)
        assert (adv.kernel_name(), dense.kernel_name()) == ("generic_rader", "generic_lds")
        for inp, eq in ((x, None), (xe, feq)):
            ref, st = R.advanced_receive(inp, nt, M, K, 2, smap, R.qpsk_points(), rounds, f_eq=eq, do_phase_compensation=pc,
                                         kind="qpsk" if dec == "auto" else "nearest", return_stages=True)
            keep = guarded(st, smap, K, M)
            assert keep.sum() >= B - 1
            got = adv.demodulate(inp) if eq is None else adv.demodulate_equalize(inp, eq)
            den = dense.demodulate(inp) if eq is None else dense.demodulate_equalize(inp, eq)
            check_err("rader_ic%d_%s_pc%d" % (rounds, dec, pc), rel_err(got[keep], ref[keep]), TOL)
            assert rel_err(got[keep], den[keep]) < 5e-6
    # what the Rader kernels do not serve: frames in / demapped symbols out -- same handle kind, generic kernels
    dem = gfdm_amd.Demodulator(M, K, 2, taps)
    dem.configure_frames(N + 9, 6, smap, True)
    xx = x[:4]
    frames = np.concatenate((xx[:, -6:], xx, xx[:, :3]), axis=1)
    assert rel_err(dem.demodulate_frames(frames), R.demap_from_resources(R.demodulate(xx, nt, M, K, 2), M, K, smap, True)) < TOL


def test_ic_with_complex_asymmetric_taps_uses_general_convolution():
    """The IC rounds use the real-symmetric convolution kernel only when g = IDFT(ic)/M is real and even; arbitrary
    complex taps must take the general path and still match the oracle."""
    import gfdm_amd
    rng = np.random.default_rng(11)
    for (M, K, L) in ((9, 64, 2), (15, 128, 4), (5, 32, 2)):
        taps = rng.standard_normal(M * L) + 1j * rng.standard_normal(M * L)
        nt = R.normalize_taps(taps, M)
        B, N = 5, M * K
        d = qpsk(rng, (B, N))
        x = R.modulate(d, nt, M, K, L) + 0.05 * (rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N)))
        adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points())
        ref, st = R.advanced_receive(x, nt, M, K, L, np.arange(K), R.qpsk_points(), 2, kind="qpsk", return_stages=True)
        keep = guarded(st, np.arange(K), K, M)
        assert keep.sum() >= 3
        assert rel_err(adv.demodulate(x)[keep], ref[keep]) < TOL


@pytest.mark.parametrize("M,K,L,alpha", [(9, 64, 2, 0.2), (15, 128, 4, 0.2), (5, 32, 2, 0.5), (15, 64, 2, 0.2), (9, 128, 2, 0.2), (16, 256, 2, 0.3),
                                         (4, 16, 3, 0.4), (8, 32, 2, 0.3), (12, 16, 2, 0.35), (7, 512, 2, 0.3),
                                         (5, 16, 2, 0.4), (10, 128, 2, 0.3)])      # (the tile padding of the crossings: none for an odd M at K = 16; an even M at four groups)
def test_ic_rounds_on_the_matrix_cores_match_the_vector_alu(M, K, L, alpha):
    """QPSK decisions + a real even IC kernel run the cancellation rounds as f16 MFMAs (IcMfma, gfdm_rowlane_impl.h: decisions exact in
    f16, IC taps as a three-term f16 split); handles created under set_ic_matrix_cores(False) run the same rounds on the vector ALU in
    f32.  Both must match the float64 oracle to 1e-5 (MF and ZF input, partial subcarrier maps, 1-5 rounds, plain blocks and demapped
    frames) and each other to well below that."""
    import gfdm_amd
    rng = np.random.default_rng(5 * M + K + L)
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    nt = R.normalize_taps(taps, M)
    N, B = M * K, 9
    smap = np.concatenate((np.arange(1, K // 2 - 1), np.arange(K // 2 + 2, K)))
    for sm in (np.arange(K), smap):
        d = np.zeros((B, K, M), complex)
        d[:, sm, :] = qpsk(rng, (B, len(sm), M))
        x = R.modulate(d.reshape(B, N), nt, M, K, L) + 0.02 * (rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N)))
        feq = np.fft.fft(np.array([1, .5, .1j, .1 + .05j]), N)[None, :] * np.exp(0.01j * np.arange(B))[:, None]
        xe = np.fft.ifft(np.fft.fft(x, axis=-1) * feq, axis=-1)
        for ic_iter in (1, 2, 5):
            prev = gfdm_amd.set_ic_matrix_cores(2)         # matrix cores wherever the form applies (the default: subcarriers >= 128)
            try:
                mx = gfdm_amd.AdvancedReceiver(M, K, L, taps, sm, ic_iter, R.qpsk_points())
                gfdm_amd.set_ic_matrix_cores(0)
                va = gfdm_amd.AdvancedReceiver(M, K, L, taps, sm, ic_iter, R.qpsk_points())
            finally:
                gfdm_amd.set_ic_matrix_cores(prev)
            for inp, eq in ((x, None), (xe, feq)):
                ref, st = R.advanced_receive(inp, nt, M, K, L, sm, R.qpsk_points(), ic_iter, f_eq=eq, kind="qpsk", return_stages=True)
                keep = guarded(st, sm, K, M)
                assert keep.sum() >= 3            # (an even number of timeslots leaves small decision margins: few blocks pass the guard)
                a = mx.demodulate(inp) if eq is None else mx.demodulate_equalize(inp, eq)
                b = va.demodulate(inp) if eq is None else va.demodulate_equalize(inp, eq)
                assert rel_err(a[keep], ref[keep]) < TOL and rel_err(b[keep], ref[keep]) < TOL
                assert rel_err(a[keep], b[keep]) < 2e-6
    # frames in, demapped symbols out (the store stage reads the rows back from the tile)
    prev = gfdm_amd.set_ic_matrix_cores(2)
    try:
        mx = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 2, R.qpsk_points())
    finally:
        gfdm_amd.set_ic_matrix_cores(prev)
    for per_timeslot in (True, False):
        mx.configure_frames(N + 7, 5, smap, per_timeslot)
        frames = np.concatenate((x[:, -5:], x, x[:, :2]), axis=1)
        ref, st = R.advanced_receive(x, nt, M, K, L, smap, R.qpsk_points(), 2, kind="qpsk", return_stages=True)
        keep = guarded(st, smap, K, M)
        want = R.demap_from_resources(ref, M, K, smap, per_timeslot)
        assert rel_err(mx.demodulate_frames(frames)[keep], want[keep]) < TOL


def test_decision_rule_a_handle_runs_is_reported():
    """gfdm_hip_advanced_receiver_decision: the sign tests (and with them the matrix-core cancellation rounds) only for GNU Radio's unit QPSK / BPSK
    points -- every component within 6 * FLT_EPSILON relative, which admits GNU Radio's 0.707107 literal --; scaled, rotated or perturbed points are decided by the nearest-point rule over the points
    as given, also when 'qpsk' / 'bpsk' was asked for, and the handle says so."""
    import gfdm_amd
    M, K, L = 9, 64, 2
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    q = R.qpsk_points()
    mk = lambda pts, dec: gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, pts, decision=dec).decision_rule()
    assert mk(q, "auto") == "qpsk" and mk(q, "qpsk") == "qpsk" and mk(q, "nearest") == "nearest"
    assert mk(q.astype(np.complex64) * np.complex64(1 + 1.2e-7), "auto") == "qpsk"           # one or two ulps off: still the unit constellation
    # gr::digital::constellation_qpsk is built from the LITERAL SQRT_TWO = 0.707107 (3.8 float ulps from 1 / sqrt 2): GNU Radio's own object must get the
    # sign tests -- and the handle's results with it stay within the tolerance of the oracle run on those very points
    gr_q = (np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) * np.float32(0.707107)).astype(np.complex64)
    assert mk(gr_q, "auto") == "qpsk" and mk(gr_q, "qpsk") == "qpsk"
    rng = np.random.default_rng(5)
    nt = R.normalize_taps(taps, M)
    x = R.modulate(qpsk(rng, (6, M * K)), nt, M, K, L)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, gr_q)
    assert rel_err(adv.demodulate(x), R.advanced_receive(x, nt, M, K, L, np.arange(K), gr_q.astype(complex), 2, kind="qpsk")) < TOL
    assert mk(q * (1 + 1e-6), "auto") == "nearest" and mk(q * (1 + 1e-6), "qpsk") == "nearest"
    assert mk(2 * q, "qpsk") == "nearest" and mk(q * np.exp(0.3j), "qpsk") == "nearest"
    assert mk(np.array([-1, 1]), "auto") == "bpsk" and mk(np.array([-2, 2]), "bpsk") == "nearest"
    with pytest.raises(ValueError):
        mk(q[:3], "qpsk")                                                                     # rule and number of points do not match


def test_phase_compensation_removes_a_common_phase():
    """The known answer of tests/test_oracle.py::test_phase_compensation_removes_a_common_phase on the HIP path
    (lib/advanced_receiver_kernel_cc.cc:59-71,78-91): with phase compensation the receiver output does not depend on a common phase of
    the input, without it the phase stays in the output; both also against the oracle."""
    import gfdm_amd
    from test_oracle import phase_case
    for (M, K, L, alpha) in ((9, 64, 2, 0.2), (15, 128, 4, 0.2), (5, 32, 2, 0.5), (127, 16, 2, 0.3)):     # (M = 127: generic family, rounds on the matrix cores)
        nt, smap, x, keep = phase_case(M, K, L, alpha)
        B = x.shape[0]
        taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
        pc = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 2, R.qpsk_points(), do_phase_compensation=1)
        nopc = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 2, R.qpsk_points())
        clean = pc.demodulate(x)
        ref = R.advanced_receive(x, nt, M, K, L, smap, R.qpsk_points(), 2, do_phase_compensation=1, kind="qpsk")
        check_err("phase_comp_clean_%d_%d" % (M, K), rel_err(clean[keep], ref[keep]), TOL)
        for phi0 in (0.05, -0.03):
            xr = x * np.exp(1j * phi0)
            check_err("phase_comp_invariance_%d_%d" % (M, K), rel_err(pc.demodulate(xr)[keep], clean[keep]), 2e-6)
            act = lambda v: v.reshape(B, K, M)[:, smap, :]
            assert abs(np.angle(np.sum(act(nopc.demodulate(xr)) * np.conj(act(nopc.demodulate(x))))) - phi0) < 0.01


def test_duplicate_subcarrier_map_entry_counts_twice_in_the_phase_mean():
    """Known answer for lib/advanced_receiver_kernel_cc.cc:78-91, 109-123 (no Python model, no reference test): calculate_phase_offset
    iterates the subcarrier MAP, not the set of active subcarriers, and divides by map.size() * timeslots -- a subcarrier listed twice
    weighs twice in the phase mean.  The measured phase is read back from the product alone: with one IC round, phase compensation turns
    S by phi and nothing else, so out_pc - out_nopc = d0 (exp(j phi) - 1) with d0 the plain demodulator output.  Subcarriers k1 and k2
    carry symbols turned by different angles, so phi[k1] != phi[k2], and the answers are
        phi[k1, k2] = (phi[k1] + phi[k2]) / 2,   phi[k1, k2, k2] = phi[k2, k1, k2] = (phi[k1] + 2 phi[k2]) / 3.
    The oracle (restated from the same lines, read against the source) is compared on the duplicated map as well."""
    import gfdm_amd
    for (M, K, L, alpha) in ((9, 64, 2, 0.2), (15, 128, 4, 0.2), (5, 32, 2, 0.5), (127, 16, 2, 0.3)):       # (M = 127: generic family)
        rng = np.random.default_rng(7 * M + K)
        taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
        nt = R.normalize_taps(taps, M)
        N, B, k1, k2 = M * K, 4, 3, 7
        d = np.zeros((B, K, M), complex)
        d[:, k1, :] = qpsk(rng, (B, M)) * np.exp(-0.02j)
        d[:, k2, :] = qpsk(rng, (B, M)) * np.exp(0.08j)
        x = R.modulate(d.reshape(B, N), nt, M, K, L)
        d0 = gfdm_amd.Demodulator(M, K, L, taps).demodulate(x).astype(np.complex128)

        def run(smap, pc):
            return gfdm_amd.AdvancedReceiver(M, K, L, taps, np.asarray(smap), 1, R.qpsk_points(), do_phase_compensation=pc).demodulate(x)

        def phi(smap):
            diff = run(smap, 1).astype(np.complex128) - run(smap, 0)
            return np.angle(1.0 + np.sum(np.conj(d0) * diff, axis=-1) / np.sum(np.abs(d0) ** 2, axis=-1))

        p1, p2 = phi([k1]), phi([k2])
        assert np.min(np.abs(p1 - p2)) > 0.05                      # the two subcarriers do measure different offsets
        assert np.max(np.abs(phi([k1, k2]) - (p1 + p2) / 2)) < 2e-5
        assert np.max(np.abs(phi([k1, k2, k2]) - (p1 + 2 * p2) / 3)) < 2e-5
        assert np.max(np.abs(phi([k2, k1, k2]) - (p1 + 2 * p2) / 3)) < 2e-5
        ref = R.advanced_receive(x, nt, M, K, L, [k1, k2, k2], R.qpsk_points(), 1, do_phase_compensation=1, kind="qpsk")
        check_err("phase_comp_duplicate_map_%d_%d" % (M, K), rel_err(run([k1, k2, k2], 1), ref), TOL)


def test_sharded_batch_on_the_gpu():
    """The batched-blocks multi-GPU mode through its two product entry points on the ONE GPU of the test box (each device ordinal may
    be listed several times: one handle + one stream per entry): gfdm_amd.sharding.ShardedBatch (what bench.py runs, one process per
    GPU there) and the C++ template gr::gfdm::sharded_batch<Kernel> (one host thread per shard).  The shards together must reproduce
    the single-handle result bit for bit -- blocks are independent, no payload is exchanged."""
    import torch
    import gfdm_amd
    import gfdm_python  # noqa: F401
    import gfdm_testing as T
    from gfdm_amd import sharding, synth
    M, K, L = 9, 64, 2
    N, total = M * K, 1001
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    dev = torch.device("cuda:0")
    sym = synth.qpsk_symbols(0, total, N, dev)
    feq = synth.channel_response(0, total, N, dev)
    mod = gfdm_amd.Modulator(M, K, L, taps)
    xe = synth.through_channel(mod.modulate(sym), feq)
    whole = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points()).demodulate_equalize(xe, feq)
    torch.cuda.synchronize()
    # (a) Python: three shards on device 0, device-resident shard tensors, launches prepared once and replayed
    sb = sharding.ShardedBatch(lambda d: gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points(), device=d), [0, 0, 0])
    plan = sb.plan(total)
    assert [n for _, _, n in plan] == [334, 334, 333] and sb.local_blocks(total) == total
    ins = [(xe[s:s + n].contiguous(), feq[s:s + n].contiguous()) for _, s, n in plan]
    outs = [torch.empty(n, N, dtype=torch.complex64, device=dev) for _, _, n in plan]
    go = sb.prepare(gfdm_amd.lib().gfdm_hip_advanced_receiver_work_device, outs, ins, [n for _, _, n in plan])
    go()
    sb.synchronize()
    assert torch.equal(torch.cat(outs), whole)
    res = sb.run("demodulate_equalize", total, ins)
    sb.synchronize()
    assert torch.equal(torch.cat(res), whole)
    parts = sb.run_global("demodulate_equalize", [xe.cpu().numpy(), feq.cpu().numpy()], [N, N])
    assert [(s, n) for s, n, _ in parts] == [(s, n) for _, s, n in plan]
    assert np.array_equal(np.concatenate([p for _, _, p in parts]), whole.cpu().numpy())
    # (b) C++: gr::gfdm::sharded_batch over host batches, through the test-only module
    x_h, feq_h, sym_h = xe.cpu().numpy(), feq.cpu().numpy(), sym.cpu().numpy()
    got = T.sharded_advanced_receive(M, K, L, list(taps.astype(np.complex64)), list(range(K)), 2, [0, 0, 0, 0], x_h, feq_h)
    assert np.array_equal(got.reshape(total, N), whole.cpu().numpy())
    frames, shards = T.sharded_modulate(M, K, L, list(taps.astype(np.complex64)), [0, 0], sym_h)
    assert shards == [(0, 501), (501, 500)]
    assert np.array_equal(frames.reshape(total, N), mod.modulate(sym).cpu().numpy())
    assert T.shard_range(total, 2, 3) == sharding.shard_range(total, 2, 3) and T.default_device_is_per_thread()
    with pytest.raises(RuntimeError):
        T.sharded_modulate(M, K, L, list(taps.astype(np.complex64)), [0, 99], sym_h)          # no such device


def test_all_zero_and_tie_inputs_follow_the_reference_decision_rule():
    """decision_maker is '> 0': an exactly-zero component maps to the NEGATIVE point."""
    import gfdm_amd
    # vector-ALU rounds (compare + select) and matrix-core rounds (v_med3_f32 on x * inf: NaN for +-0 selects the negative point)
    for (M, K, L, mode) in ((9, 64, 2, 0), (9, 64, 2, 2), (15, 128, 4, 1), (15, 128, 4, 0), (5, 32, 2, 2)):
        prev = gfdm_amd.set_ic_matrix_cores(mode)
        try:
            taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
            nt = R.normalize_taps(taps, M)
            adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points())
        finally:
            gfdm_amd.set_ic_matrix_cores(prev)
        x = np.zeros((3, M * K), np.complex64)
        x[1] = -0.0 - 0.0j                        # negative zeros as well
        x[2, ::7] = 1e-42                         # ... and a few subnormal samples
        ref = R.advanced_receive(x[:2], nt, M, K, L, np.arange(K), R.qpsk_points(), 2, kind="qpsk")
        got = adv.demodulate(x)
        assert np.abs(ref).max() > 5e-4            # the all-negative decisions leave a non-zero cancellation term
        assert rel_err(got[:2], ref) < TOL and np.all(np.isfinite(got))


def test_device_entry_points_are_graph_capturable():
    """include/gfdm_hip.h promises that *_device calls only enqueue work (no allocation, no synchronisation): capture a
    modulate -> demodulate -> IC chain into a HIP graph, replay it on new input, compare with the eager result."""
    import torch
    import gfdm_amd
    from gfdm_amd import synth
    dev = torch.device("cuda:0")
    M, K, L, B = 9, 64, 2, 300
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    mod, dem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points())
    sym = synth.qpsk_symbols(0, B, M * K, dev)
    x, y, z = (torch.empty_like(sym) for _ in range(3))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up outside capture
        mod.modulate(sym, out=x); dem.demodulate(x, out=y); adv.demodulate(x, out=z)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        mod.modulate(sym, out=x)
        dem.demodulate(x, out=y)
        adv.demodulate(x, out=z)
    sym.copy_(synth.qpsk_symbols(777, B, M * K, dev))   # new input, same buffers
    graph.replay()
    torch.cuda.synchronize()
    nt = R.normalize_taps(taps, M)
    s_h = sym.cpu().numpy()
    ref_x = R.modulate(s_h, nt, M, K, L)
    assert rel_err(x.cpu().numpy(), ref_x) < TOL
    assert rel_err(y.cpu().numpy(), R.demodulate(ref_x, nt, M, K, L)) < TOL
    assert float((z - sym).abs().max()) < 0.2


@pytest.mark.parametrize("M,K,L", [(3, 7, 2), (5, 13, 2), (4, 28, 3), (7, 30, 2), (3, 36, 4), (9, 50, 2), (5, 66, 2), (2, 99, 2), (11, 8, 2), (6, 40, 5),
                                   (1, 16, 2), (3, 2, 2)])
def test_generic_family_any_subcarrier_count(M, K, L):
    """The generic family's subcarrier transform is a mixed-radix Stockham FFT (radices 4, 2, 3, 5, 7, 11, 13 here, a prime K as one
    direct pass): modulator, ZF receiver and IC receiver against the float64 oracle on awkward shapes."""
    import gfdm_amd
    rng = np.random.default_rng(100 * K + 10 * M + L)
    N, B = M * K, 5
    taps = get_frequency_domain_filter("rrc", 0.4, M, K, L)
    nt = R.normalize_taps(taps, M)
    with gfdm_amd.generic_family_for_testing():                  # (power-of-two K would otherwise be instantiated at run time)
        mod, dem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
    assert mod.kernel_name() == "generic_lds"
    sym = qpsk(rng, (B, N))
    x = R.modulate(sym, nt, M, K, L)
    assert rel_err(mod.modulate(sym), x) < TOL
    feq = np.fft.fft(np.array([1, .3 - .2j, .1j]), N)[None, :] * np.ones((B, 1))
    xe = np.fft.ifft(np.fft.fft(x, axis=-1) * feq, axis=-1)
    assert rel_err(dem.demodulate_equalize(xe, feq), R.demodulate(xe, nt, M, K, L, feq)) < TOL
    assert rel_err(dem.demodulate(x), R.demodulate(x, nt, M, K, L)) < TOL
    with gfdm_amd.generic_family_for_testing():
        adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points())
    ref, st = R.advanced_receive(xe, nt, M, K, L, np.arange(K), R.qpsk_points(), 2, f_eq=feq, kind="qpsk", return_stages=True)
    keep = guarded(st, np.arange(K), K, M)
    if keep.any():
        assert rel_err(adv.demodulate_equalize(xe, feq)[keep], ref[keep]) < TOL


@pytest.mark.parametrize("M,K,L", [(127, 16, 2), (32, 16, 2), (33, 20, 2), (48, 6, 3), (63, 37, 2), (64, 8, 2), (100, 16, 4), (127, 40, 2), (255, 3, 2),
                                   (35, 74, 2)])
def test_generic_family_matrix_core_timeslot_transforms(M, K, L):
    """From 32 timeslots on the generic family runs its timeslot transforms on the matrix cores (mx_dft, gfdm_generic.hip: constant cosine / sine
    matrices x the block's sample pairs, v_mfma_f32_16x16x4_f32 -- f32 operands and sums).  Modulator, MF / ZF receiver, IC receiver and the stand-alone
    transform_subcarriers_to_td / cancel_sc_interference against the float64 oracle, and the vector-ALU form of the same handle kind beside it
    (odd, even, prime M; row counts that are not a multiple of the 16-row operand group; several row groups per LDS chunk and one)."""
    import gfdm_amd
    rng = np.random.default_rng(7 * K + 3 * M + L)
    N, B = M * K, 4
    taps = get_frequency_domain_filter("rrc", 0.35, M, K, L)
    nt = R.normalize_taps(taps, M)
    sym = qpsk(rng, (B, N))
    x = R.modulate(sym, nt, M, K, L)
    feq = np.fft.fft(np.array([1, .25 - .2j, .1j]), N)[None, :] * np.ones((B, 1))
    xe = np.fft.ifft(np.fft.fft(x, axis=-1) * feq, axis=-1)
    ref, st = R.advanced_receive(xe, nt, M, K, L, np.arange(K), R.qpsk_points(), 2, f_eq=feq, kind="qpsk", return_stages=True)
    keep = guarded(st, np.arange(K), K, M)
    outs = {}
    for on in (2, 0):                                            # 2: the matrix-core form wherever it fits, also where mode 1 would not choose it
        prev = gfdm_amd.set_dft_matrix_cores(on)
        try:
            with gfdm_amd.generic_family_for_testing():
                mod, dem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
                adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points())
        finally:
            gfdm_amd.set_dft_matrix_cores(prev)
        assert mod.kernel_name() == "generic_lds"
        o = outs[on] = dict(mod=mod.modulate(sym), mf=dem.demodulate(x), zf=dem.demodulate_equalize(xe, feq), ic=adv.demodulate_equalize(xe, feq))
        check_err("generic_mx%d_mod" % on, rel_err(o["mod"], x), TOL)
        check_err("generic_mx%d_mf" % on, rel_err(o["mf"], R.demodulate(x, nt, M, K, L)), TOL)
        check_err("generic_mx%d_zf" % on, rel_err(o["zf"], R.demodulate(xe, nt, M, K, L, feq)), TOL)
        if keep.any():
            check_err("generic_mx%d_ic" % on, rel_err(o["ic"][keep], ref[keep]), TOL)
        fd = dem.fft_filter_downsample(x)
        td = dem.transform_subcarriers_to_td(fd)
        check_err("generic_mx%d_to_td" % on, rel_err(td, R.transform_subcarriers_to_td(R.fft_filter_downsample(x, nt, M, K, L), M, K)), TOL)
        check_err("generic_mx%d_cancel" % on, rel_err(dem.cancel_sc_interference(sym, fd),
                                                      R.cancel_sc_interference(sym, np.asarray(fd), R.ic_filter_taps(nt, M, L), M, K)), TOL)
    # the two forms compute the same sums in a different order
    assert rel_err(outs[2]["mod"], outs[0]["mod"]) < 2e-6
    assert rel_err(outs[2]["mf"], outs[0]["mf"]) < 2e-6


@pytest.mark.parametrize("M,K,L,alpha", [(7, 16, 2, 0.3), (13, 32, 4, 0.4), (11, 8, 2, 0.5), (27, 128, 2, 0.2), (6, 256, 2, 0.3), (28, 64, 2, 0.1), (5, 4, 8, 0.5),
                                         (10, 96, 2, 0.35), (21, 12, 2, 0.35), (9, 48, 4, 0.3), (7, 240, 2, 0.2), (15, 80, 2, 0.3), (9, 15, 2, 0.4), (3, 6, 2, 0.5),
                                         (4, 100, 2, 0.5), (5, 20, 6, 0.4), (9, 512, 2, 0.3), (15, 1024, 2, 0.2),
                                         (9, 384, 2, 0.3), (5, 600, 2, 0.2), (9, 200, 2, 0.4), (37, 32, 2, 0.3),
                                         (9, 34, 2, 0.3), (5, 93, 2, 0.4), (3, 589, 2, 0.2), (9, 31, 2, 0.3)])
def test_row_lane_kernels_instantiated_at_run_time(M, K, L, alpha, tmp_path, monkeypatch):
    """Shapes outside the compiled list get the row-lane kernels instantiated through hiprtc when the handle is created
    (gfdm_jit.hip) -- K a power of two, or K = R0 x R1 with both factors <= 16 (96 = 6 x 16, 12, 48 = 3 x 16, 240 = 15 x 16, 80, 15,
    6, 100 = 10 x 10, 20; 512, 1024 and the K without a two-factor plan -- 384, 600, 200 -- take three wide passes; 34 = 2 x 17, 93 = 3 x 31,
    589 = 19 x 31 and the prime 31 use butterflies of up to 32 points): every mode against the oracle, and the switch that turns the run-time instantiation off."""
    import gfdm_amd
    monkeypatch.setenv("GFDM_HIP_CACHE_DIR", str(tmp_path))       # cold cache: really compile
    rng = np.random.default_rng(31 * M + K + L)
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    nt = R.normalize_taps(taps, M)
    N, B = M * K, 37
    smap = np.arange(K) if K < 8 else np.concatenate((np.arange(1, K // 2 - 1), np.arange(K // 2 + 1, K)))
    built = lambda: len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")])
    mod = gfdm_amd.Modulator(M, K, L, taps)
    n_mod = built()
    dem = gfdm_amd.Demodulator(M, K, L, taps)
    n_dem = built()
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 2, R.qpsk_points())
    assert (mod.kernel_name(), dem.kernel_name(), adv.kernel_name()) == ("rowlane_jit",) * 3
    if (M, K, L) == (10, 96, 2):          # (a shape no other test of this process has loaded) every handle compiles only its own kernels
        assert (n_mod, n_dem, built()) == (1, 2, 3)
    prev = gfdm_amd.set_jit(gfdm_amd.JIT_OFF)
    try:
        assert prev == gfdm_amd.JIT_IN_CONSTRUCTOR and gfdm_amd.Demodulator(M, K, L, taps).kernel_name() == "generic_lds"
    finally:
        gfdm_amd.set_jit(prev)
    d = np.zeros((B, K, M), complex)
    d[:, smap, :] = qpsk(rng, (B, len(smap), M))
    d = d.reshape(B, N)
    x = R.modulate(d, nt, M, K, L)
    feq = np.fft.fft(np.array([1, .5, .1j, .1 + .05j]), N)[None, :] * np.exp(0.01j * np.arange(B))[:, None]
    xe = np.fft.ifft(np.fft.fft(x, axis=-1) * feq, axis=-1)
    S = R.fft_filter_downsample(x, nt, M, K, L)
    assert rel_err(mod.modulate(d), x) < TOL
    assert rel_err(dem.fft_filter_downsample(x), S) < TOL
    assert rel_err(dem.fft_equalize_filter_downsample(xe, feq), R.fft_filter_downsample(xe, nt, M, K, L, feq)) < TOL
    assert rel_err(dem.demodulate(x), R.demodulate(x, nt, M, K, L)) < TOL
    assert rel_err(dem.demodulate_equalize(xe, feq), R.demodulate(xe, nt, M, K, L, feq)) < TOL
    for inp, eq in ((x, None), (xe, feq)):
        ref, st = R.advanced_receive(inp, nt, M, K, L, smap, R.qpsk_points(), 2, f_eq=eq, kind="qpsk", return_stages=True)
        keep = guarded(st, smap, K, M)
        got = adv.demodulate(inp) if eq is None else adv.demodulate_equalize(inp, eq)
        assert keep.sum() >= B // 2 and rel_err(got[keep], ref[keep]) < TOL
    # frames in, demapped symbols out (load offset / gather store of the same kernels)
    dem.configure_frames(N + 12, 5, smap, True)
    frames = rng.standard_normal((B, N + 12)) + 1j * rng.standard_normal((B, N + 12))
    frames[:, 5:5 + N] = xe
    assert rel_err(dem.demodulate_frames(frames, feq), R.demap_from_resources(R.demodulate(xe, nt, M, K, L, feq), M, K, smap, True)) < TOL


def test_run_time_instantiation_off_the_constructors_critical_path(tmp_path, monkeypatch):
    """gfdm_hip_set_jit modes 2 / 3 and gfdm_hip_precompile (include/gfdm_hip.h): with a cold cache a handle for a many-timeslot shape
    comes back at once on the generic family, gives the right answer there, and switches to the tuned kernels once the background
    build is done; after that (code objects cached) the constructor takes the tuned kernels directly; a precompiled shape starts on
    them even in its first process."""
    import time
    import gfdm_amd
    monkeypatch.setenv("GFDM_HIP_CACHE_DIR", str(tmp_path))
    M, K, L = 19, 32, 2
    taps = get_frequency_domain_filter("rrc", 0.3, M, K, L)
    nt = R.normalize_taps(taps, M)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((5, M * K)) + 1j * rng.standard_normal((5, M * K))
    ref = R.demodulate(x, nt, M, K, L)
    prev = gfdm_amd.set_jit(gfdm_amd.JIT_AUTO)
    try:
        t0 = time.perf_counter()
        dem = gfdm_amd.Demodulator(M, K, L, taps)                       # 19 timeslots, nothing cached: background build
        t_create = time.perf_counter() - t0
        assert t_create < 1.0 and dem.kernel_name() == "generic_lds"
        assert rel_err(dem.demodulate(x), ref) < TOL                    # served by the generic family meanwhile
        deadline = time.perf_counter() + 300
        while dem.kernel_name() != "rowlane_jit" and time.perf_counter() < deadline:
            time.sleep(0.25)
        assert dem.kernel_name() == "rowlane_jit"
        assert rel_err(dem.demodulate(x), ref) < TOL                    # ... and by the tuned kernels afterwards
        t0 = time.perf_counter()
        dem2 = gfdm_amd.Demodulator(M, K, L, taps)                      # cached now: tuned kernels straight away
        assert dem2.kernel_name() == "rowlane_jit" and time.perf_counter() - t0 < 2.0
        # a quick shape (timeslots <= 16) is compiled inside the constructor under JIT_AUTO
        assert gfdm_amd.Demodulator(6, 32, 2, get_frequency_domain_filter("rrc", 0.3, 6, 32, 2)).kernel_name() == "rowlane_jit"
        # precompile: the deployment step -- afterwards the first handle of the shape starts on the tuned kernels
        M2 = 18
        gfdm_amd.precompile(M2, K, L, 1 | 8)                            # receive + modulate parts
        taps2 = get_frequency_domain_filter("rrc", 0.3, M2, K, L)
        t0 = time.perf_counter()
        d3, m3 = gfdm_amd.Demodulator(M2, K, L, taps2), gfdm_amd.Modulator(M2, K, L, taps2)
        assert (d3.kernel_name(), m3.kernel_name()) == ("rowlane_jit", "rowlane_jit") and time.perf_counter() - t0 < 2.0
        gfdm_amd.precompile(9, 64, 2)                                   # compiled into the library: nothing to do, no error
        with pytest.raises(gfdm_amd.GfdmHipError):
            gfdm_amd.precompile(127, 16, 2)                             # generic family only
        # the channel estimator handle follows the same policy (its kernel is the fifth part of a shape)
        Me, Ke, Ae = 17, 32, 24
        pre = np.tile(np.fft.ifft(np.exp(2j * np.pi * rng.random(Ke))) * np.sqrt(Ke), 2)
        rx_pre = (np.tile(pre, (3, 1)) * np.exp(0.3j * np.arange(3))[:, None]).astype(np.complex64)
        est = gfdm_amd.ChannelEstimator(Me, Ke, Ae, True, 1, pre)
        assert est.kernel_name() == "generic_lds"
        ref_e = R.estimate_frame(rx_pre, pre.astype(np.complex64), Me, Ke, Ae, True)
        assert rel_err(est.estimate_frame(rx_pre), ref_e) < TOL
        deadline = time.perf_counter() + 300
        while est.kernel_name() != "rowlane_jit" and time.perf_counter() < deadline:
            time.sleep(0.25)
        assert est.kernel_name() == "rowlane_jit" and rel_err(est.estimate_frame(rx_pre), ref_e) < TOL
        # JIT_BACKGROUND for any shape, also a quick one
        gfdm_amd.set_jit(gfdm_amd.JIT_BACKGROUND)
        mod = gfdm_amd.Modulator(7, 8, 2, get_frequency_domain_filter("rrc", 0.3, 7, 8, 2))
        assert mod.kernel_name() in ("generic_lds", "rowlane_jit")
        deadline = time.perf_counter() + 120
        while mod.kernel_name() != "rowlane_jit" and time.perf_counter() < deadline:
            time.sleep(0.1)
        assert mod.kernel_name() == "rowlane_jit"
    finally:
        gfdm_amd.set_jit(prev)


def test_handles_on_concurrent_host_threads():
    """GNU Radio runs every block's work() on its own thread: a modulator, two receivers and an estimator, each with its OWN handle,
    working at the same time from different host threads (host-pointer entry points, ctypes drops the GIL) must give exactly the
    results of the same calls made one after the other.  Handles are not shared between threads (boundary contract)."""
    import threading
    import gfdm_amd
    rng = np.random.default_rng(77)
    M, K, L = 9, 64, 2
    N = M * K
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    pre = (rng.standard_normal(2 * K) + 1j * rng.standard_normal(2 * K)) / np.sqrt(2)
    sym = qpsk(rng, (40, N)).astype(np.complex64)
    feq = (np.fft.fft(np.array([1, .5, .1j]), N)[None, :] * np.ones((40, 1))).astype(np.complex64)
    rxp = (rng.standard_normal((40, 2 * K)) + 1j * rng.standard_normal((40, 2 * K))).astype(np.complex64)

    def jobs():
        mod, dem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
        adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points())
        est = gfdm_amd.ChannelEstimator(M, K, 52, True, 1, pre)
        small = gfdm_amd.Demodulator(5, 12, 2, get_frequency_domain_filter("rrc", 0.3, 5, 12, 2))     # a run-time instantiated shape
        x = mod.modulate(sym)
        return [lambda: mod.modulate(sym), lambda: dem.demodulate(x), lambda: adv.demodulate_equalize(x, feq),
                lambda: est.estimate_frame(rxp), lambda: small.demodulate(sym[:, :60])]

    serial = [f() for f in jobs()]
    fns = jobs()
    results, errors = [None] * len(fns), []

    def worker(i):
        try:
            out = None
            for _ in range(25):                       # single blocks and whole batches interleave on the GPU
                out = fns[i]()
            results[i] = out
        except Exception as e:                        # noqa: BLE001 - reported below
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(fns))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for a, b in zip(serial, results):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("M,K,L", [(15, 1040, 2), (49, 512, 2), (5, 4096, 4), (127, 100, 2)])
def test_blocks_larger_than_the_lds(M, K, L):
    """Blocks whose tiles do not fit the 160 KiB of a CU (N > ~10 000: K = 1024 x M = 15 ...) run the generic kernels with their tiles
    in a global scratch buffer (K = 1024 itself is a row-lane shape: three wide passes): every entry point against the oracle, frames / demapper, the fused transmitter, and a batch long
    enough to go down in several chunks of the scratch buffer."""
    import torch
    import gfdm_amd
    rng = np.random.default_rng(M + K)
    N, B = M * K, 3
    taps = get_frequency_domain_filter("rrc", 0.3, M, K, L)
    nt = R.normalize_taps(taps, M)
    A = K - K // 8
    smap = np.concatenate((np.arange(1, A // 2 + 1), np.arange(K - A // 2, K)))
    mod, dem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 2, R.qpsk_points())
    assert (mod.kernel_name(), dem.kernel_name(), adv.kernel_name()) == ("generic_lds",) * 3
    d = np.zeros((B, K, M), complex)
    d[:, smap, :] = qpsk(rng, (B, A, M))
    d = d.reshape(B, N)
    x = R.modulate(d, nt, M, K, L)
    assert rel_err(mod.modulate(d), x) < TOL
    feq = np.fft.fft(np.array([1, .4 - .2j, .1j]), N)[None, :] * np.exp(0.02j * np.arange(B))[:, None]
    xe = np.fft.ifft(np.fft.fft(x, axis=-1) * feq, axis=-1)
    S = R.fft_filter_downsample(x, nt, M, K, L)
    assert rel_err(dem.fft_filter_downsample(x), S) < TOL
    assert rel_err(dem.demodulate(x), R.demodulate(x, nt, M, K, L)) < TOL
    assert rel_err(dem.demodulate_equalize(xe, feq), R.demodulate(xe, nt, M, K, L, feq)) < TOL
    td = R.transform_subcarriers_to_td(S, M, K)
    assert rel_err(dem.transform_subcarriers_to_td(S), td) < TOL
    assert rel_err(dem.cancel_sc_interference(d, S), R.cancel_sc_interference(d, S, R.ic_filter_taps(nt, M, L), M, K)) < TOL
    ref, st = R.advanced_receive(xe, nt, M, K, L, smap, R.qpsk_points(), 2, f_eq=feq, kind="qpsk", return_stages=True)
    keep = guarded(st, smap, K, M)
    got = adv.demodulate_equalize(xe, feq)
    assert keep.any() and rel_err(got[keep], ref[keep]) < TOL
    # frames in, demapped symbols out, with IC (the combination that has no LDS form even for mid-sized blocks)
    adv.configure_frames(N + 9, 4, smap, True)
    frames = rng.standard_normal((B, N + 9)) + 1j * rng.standard_normal((B, N + 9))
    frames[:, 4:4 + N] = xe
    assert rel_err(adv.demodulate_frames(frames, feq)[keep], R.demap_from_resources(ref, M, K, smap, True)[keep]) < TOL
    # fused transmitter (mapper in front, prefix + preamble behind)
    window = np.ones(N + 24 + 8, complex)
    pre = rng.standard_normal(16) + 1j * rng.standard_normal(16)
    tx = gfdm_amd.Transmitter(M, K, A, 24, 8, 0, smap, True, L, taps, window, [0, 5], [pre, pre])
    sym = qpsk(rng, (B, A * M))
    outs = tx.transmit(sym)
    for port, s in enumerate((0, 5)):
        assert rel_err(outs[port], R.transmit(sym, nt, M, K, L, smap, True, 24, 8, 0, window, s, pre)) < TOL
    # a batch that needs several chunks of the scratch buffer: five distinct blocks repeated, outputs must repeat exactly
    nb = int((256 << 20) // (3 * N * 8)) + 70
    dev = torch.device("cuda:0")
    base = torch.tensor(xe[:3].astype(np.complex64), device=dev)
    big = base.repeat((nb + 2) // 3, 1)[:nb].contiguous()
    big_eq = torch.tensor(feq[:3].astype(np.complex64), device=dev).repeat((nb + 2) // 3, 1)[:nb].contiguous()
    adv2 = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 2, R.qpsk_points())
    out = adv2.demodulate_equalize(big, big_eq)
    torch.cuda.synchronize()
    first = out[:3]
    for b in (3, nb // 2 // 3 * 3, (nb - 3) // 3 * 3):
        assert torch.equal(out[b:b + 3], first)
    assert rel_err(first.cpu().numpy()[keep], ref[keep]) < TOL


def test_process_exit_with_background_builds_in_flight(tmp_path):
    """A process that ends while the kernels of new shapes are still being instantiated in the background (gfdm_hip_set_jit modes 2 / 3: a flowgraph
    torn down right after it was built) must exit cleanly and soon: the builds run on a pool of two threads fed from a queue, gfdm_hip_quiesce --
    registered with atexit by both Python surfaces, and the library's own unload path -- drops what is queued and lets the builds in flight finish
    without touching the GPU.  (Before the pool existed every shape started a thread of its own and an interpreter exit with dozens of compiles in
    flight ended in a segmentation fault.)  Eight uncached shapes, every handle answers from the generic family meanwhile."""
    import subprocess
    import sys
    import time
    code = r"""
import os, sys, time
sys.path.insert(0, os.path.join(%r, "gr-gfdm_amd", "python")); sys.path.insert(0, os.path.join(%r, "gr-gfdm_amd", "lib")); sys.path.insert(0, os.path.join(%r, "oracle"))
import numpy as np
import gfdm_amd, gfdm_python
import gfdm_ref as R
from gfdm_amd.filters import get_frequency_domain_filter
gfdm_amd.set_jit(gfdm_amd.JIT_BACKGROUND)
rng = np.random.default_rng(5)
keep = []
for i, (M, K) in enumerate(((17, 16), (18, 32), (19, 16), (20, 8), (21, 32), (22, 16), (23, 8), (25, 16))):
    taps = get_frequency_domain_filter("rrc", 0.3, M, K, 2)
    t0 = time.time()
    dem = gfdm_amd.Demodulator(M, K, 2, taps) if i %% 2 else gfdm_python.Demodulator(M, K, 2, list(taps.astype(np.complex64)))
    assert time.time() - t0 < 2.0, "constructor waited for a compile"
    x = (rng.standard_normal(M * K) + 1j * rng.standard_normal(M * K)).astype(np.complex64)
    y = np.asarray(dem.demodulate(x)).ravel()
    ref = R.demodulate(x[None, :], R.normalize_taps(taps, M), M, K, 2)[0]
    assert np.linalg.norm(y - ref) / np.linalg.norm(ref) < 1e-5
    keep.append(dem)
print("built", len(keep), flush=True)
""" % ((os.path.dirname(os.path.dirname(os.path.abspath(__file__))),) * 3)
    env = dict(os.environ, GFDM_HIP_CACHE_DIR=str(tmp_path))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert "built 8" in r.stdout
    assert time.time() - t0 < 300                                  # two builds in flight at most (10-70 s each), the other six dropped


def test_damaged_cache_entry_is_recompiled(tmp_path, monkeypatch):
    """a truncated code object in the disk cache of the run-time instantiated kernels must not strand the shape on the generic family:
    the part is compiled afresh and the cache entry replaced"""
    import gfdm_amd
    monkeypatch.setenv("GFDM_HIP_CACHE_DIR", str(tmp_path))
    M, K, L = 3, 40, 2                                           # a shape no other test loads in this process
    L_ = gfdm_amd.lib()
    assert L_.gfdm_hip_jit_build_for_testing(M, K, L, 3) == 0    # fills the cache without loading the module
    files = [f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]
    assert len(files) == 1
    path = os.path.join(tmp_path, files[0])
    good = os.path.getsize(path)
    with open(path, "r+b") as f:
        f.truncate(good // 3)
    taps = get_frequency_domain_filter("rrc", 0.3, M, K, L)
    mod = gfdm_amd.Modulator(M, K, L, taps)
    assert mod.kernel_name() == "rowlane_jit" and os.path.getsize(path) == good
    rng = np.random.default_rng(0)
    d = qpsk(rng, (4, M * K))
    assert rel_err(mod.modulate(d), R.modulate(d, R.normalize_taps(taps, M), M, K, L)) < TOL
