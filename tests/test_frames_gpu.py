"""GPU parity tests of the receivers on raw frames with demapped output: cyclic-prefix removal
(add_cyclic_prefix_cc::remove_cyclic_prefix) as the receiver kernel's load stage, resource demapper
(resource_mapper_kernel_cc::demap_from_resources) as its store stage."""
import numpy as np
import pytest

import gfdm_ref as R
from conftest import assert_places, have_gpu, load_tx_golden, rel_err
from gfdm_amd.filters import get_frequency_domain_filter

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    if not have_gpu():
        pytest.fail("no MI355X visible: the HIP path cannot run (there is no CPU fallback to test instead)")


def qpsk(rng, shape):
    return ((1 - 2 * rng.integers(0, 2, shape)) + 1j * (1 - 2 * rng.integers(0, 2, shape))) / np.sqrt(2)


@pytest.mark.parametrize("name", ["tx_ref_k64_m9_cdd", "tx_k32_m5_short"])
def test_frames_chain_matches_pygfdm(name):
    """frame -> (prefix removal) -> demodulate -> (demap) against the reference model's own chain (make_golden_tx.py)."""
    import gfdm_amd
    g = load_tx_golden(name)
    M, K, L = g["M"], g["K"], g["L"]
    plen = g["preambles"].shape[1]
    frames = g["pygfdm_frames"][0][:, plen:]                    # the GNU Radio flowgraph strips the preamble before the receiver
    frame_len = g["cp"] + M * K + g["cs"]
    assert frames.shape[1] == frame_len
    dem = gfdm_amd.Demodulator(M, K, L, g["taps"])
    dem.configure_frames(frame_len, g["cp"], g["smap"], True)
    got = dem.demodulate_frames(frames)
    assert got.shape == g["pygfdm_rx_symbols"].shape
    assert rel_err(got, g["pygfdm_rx_symbols"]) < TOL
    assert_places(got, g["pygfdm_rx_symbols"], 5)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, g["taps"], g["smap"], 0, R.qpsk_points())
    adv.configure_frames(frame_len, g["cp"], g["smap"], True)
    assert np.array_equal(adv.demodulate_frames(frames), got)   # zero IC rounds = plain receiver
    if g["symbols"].shape[1] == g["A"] * M:                     # every resource carries a QPSK symbol: IC recovers them
        adv.set_ic(8)
        assert np.max(np.abs(adv.demodulate_frames(frames) - g["symbols"])) < 0.2


@pytest.mark.parametrize("M,K,L,alpha", [(9, 64, 2, 0.2), (15, 128, 4, 0.2), (5, 32, 2, 0.5), (31, 256, 2, 0.1), (7, 12, 2, 0.3)])
@pytest.mark.parametrize("per_ts", [True, False])
def test_frames_against_oracle(M, K, L, alpha, per_ts):
    import gfdm_amd
    rng = np.random.default_rng(M * K + L + per_ts)
    N, B = M * K, 5
    cp, cs = max(1, K // 4), max(1, K // 8)
    frame_len = cp + N + cs + 3                                  # frames may be longer than cp + block + cs
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    nt = R.normalize_taps(taps, M)
    smap = np.sort(rng.choice(np.arange(K), size=max(2, (3 * K) // 4), replace=False))
    d = np.zeros((B, K, M), complex)
    d[:, smap, :] = qpsk(rng, (B, len(smap), M))
    block = R.modulate(d.reshape(B, N), nt, M, K, L)
    feq = np.fft.fft(np.array([1, .5, .1j, .1 + .05j]), N)[None, :] * np.exp(0.01j * np.arange(B))[:, None]
    block_ch = np.fft.ifft(np.fft.fft(block, axis=-1) * feq, axis=-1)
    frames = rng.standard_normal((B, frame_len)) + 1j * rng.standard_normal((B, frame_len))      # junk around the block
    frames[:, cp:cp + N] = block_ch
    dem = gfdm_amd.Demodulator(M, K, L, taps)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 2, R.qpsk_points())
    for h in (dem, adv):
        h.configure_frames(frame_len, cp, smap[::-1], per_ts)    # unsorted map: the reference sorts it
    x = R.remove_cyclic_prefix(frames, cp, N)
    ref_dem = R.demap_from_resources(R.demodulate(x, nt, M, K, L, feq), M, K, smap, per_ts)
    ref_adv, st = R.advanced_receive(x, nt, M, K, L, smap, R.qpsk_points(), 2, f_eq=feq, kind="qpsk", return_stages=True)
    ref_adv = R.demap_from_resources(ref_adv, M, K, smap, per_ts)
    assert rel_err(dem.demodulate_frames(frames, feq), ref_dem) < TOL
    # IC outputs only on blocks whose every decided component keeps DECISION_GUARD away from the boundary (tests/test_parity_gpu.py)
    keep = np.ones(B, bool)
    for dd in [st["d0"]] + st["iters"][:-1]:
        v = dd.reshape(-1, K, M)[:, smap, :]
        keep &= np.minimum(np.abs(v.real), np.abs(v.imag)).reshape(B, -1).min(axis=1) > 1e-4
    assert keep.sum() >= B - 1
    assert rel_err(adv.demodulate_frames(frames, feq)[keep], ref_adv[keep]) < TOL
    nshort = len(smap) * M - 5                                   # truncated output (noutput_size < active * timeslots)
    assert rel_err(dem.demodulate_frames(frames, feq, noutput_size=nshort), ref_dem[:, :nshort]) < TOL
    # no subcarrier map: prefix removal only, plain [k][m] blocks out
    dem.configure_frames(frame_len, cp)
    assert rel_err(dem.demodulate_frames(frames, feq), R.demodulate(x, nt, M, K, L, feq)) < TOL
    # ... and without a map the kernel writes whole blocks: a truncating noutput_size is refused, never a short buffer overrun
    with pytest.raises(ValueError, match="needs a subcarrier map"):
        dem.demodulate_frames(frames, feq, noutput_size=N - 7)
    assert dem.demodulate_frames(frames, feq, noutput_size=N).shape == (B, N)


def test_frames_device_path_and_errors():
    import torch
    import gfdm_amd
    from gfdm_amd import synth
    M, K, L, cp, cs = 9, 64, 2, 16, 8
    N = M * K
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    smap = np.concatenate((np.arange(1, 27), np.arange(38, 64)))
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, smap, 2, R.qpsk_points())
    with pytest.raises(ValueError, match="configure_frames"):
        adv._frame_len, adv._frame_nout = cp + N + cs, len(smap) * M
        adv.demodulate_frames(np.zeros(cp + N + cs, np.complex64))
    with pytest.raises(ValueError, match="frame_len"):
        adv.configure_frames(N, cp, smap)
    with pytest.raises(ValueError, match="unique"):
        adv.configure_frames(cp + N + cs, cp, [1, 1, 2])
    adv.configure_frames(cp + N + cs, cp, smap, True)
    with pytest.raises(ValueError, match="MUST not exceed"):
        adv.demodulate_frames(np.zeros(cp + N + cs, np.complex64), noutput_size=len(smap) * M + 1)
    dev = torch.device("cuda:0")
    tx = gfdm_amd.Transmitter(M, K, len(smap), cp, cs, 0, smap, True, L, taps, np.zeros(0, complex), [0], [np.zeros(0, complex)])
    B = 257
    sym = synth.qpsk_symbols(3, B, len(smap) * M, dev)
    frames = tx.transmit(sym)[0]                                  # fused transmitter -> fused receiver, device resident
    rec = adv.demodulate_frames(frames)
    torch.cuda.synchronize()
    assert rec.shape == sym.shape
    assert float((rec - sym).abs().max()) < 0.2
    assert np.array_equal(adv.demodulate_frames(frames.cpu().numpy()), rec.cpu().numpy())


def test_frames_pybind_surface():
    """gfdm_python.Demodulator / AdvancedReceiver: configure_frames + demodulate_frames (C++ classes over the C-ABI)."""
    import gfdm_python
    g = load_tx_golden("tx_ref_k64_m9_cdd")
    M, K, L = g["M"], g["K"], g["L"]
    plen = g["preambles"].shape[1]
    frames = g["pygfdm_frames"][0][:, plen:]
    frame_len = g["cp"] + M * K + g["cs"]
    dem = gfdm_python.Demodulator(M, K, L, g["taps"])
    dem.configure_frames(frame_len, g["cp"], g["smap"].tolist(), True)
    got = dem.demodulate_frames(frames)
    assert got.shape == g["pygfdm_rx_symbols"].shape and got.dtype == np.complex64
    assert_places(got, g["pygfdm_rx_symbols"], 5)
    adv = gfdm_python.AdvancedReceiver(M, K, L, g["taps"], g["smap"].tolist(), 8, gfdm_python.Constellation.qpsk(), 0)
    adv.configure_frames(frame_len, g["cp"], g["smap"].tolist(), True, M)
    assert np.max(np.abs(adv.demodulate_frames(frames) - g["symbols"])) < 0.2
    with pytest.raises(ValueError, match="unique"):
        dem.configure_frames(frame_len, g["cp"], [1, 1, 2], True)
