"""GPU parity tests of the composite transmitter (gr-gfdm transmitter_kernel =
resource mapper -> modulator -> cyclic prefix/suffix with cyclic shift + window ramp -> preamble), which runs as ONE
fused HIP kernel.  Expectations: the pygfdm golden frames (composed exactly like python/qa_transmitter_cc.py:42-55 does)
and both CPU oracles."""
import numpy as np
import pytest

import c_oracle
import gfdm_ref as R
from conftest import assert_places, have_gpu, load_tx_golden, rel_err, tx_golden_names
from gfdm_amd.filters import get_frequency_domain_filter

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    if not have_gpu():
        pytest.fail("no MI355X visible: the HIP path cannot run (there is no CPU fallback to test instead)")


def qpsk(rng, shape):
    return ((1 - 2 * rng.integers(0, 2, shape)) + 1j * (1 - 2 * rng.integers(0, 2, shape))) / np.sqrt(2)


def _tx_from_golden(g, cls):
    return cls(g["M"], g["K"], g["A"], g["cp"], g["cs"], g["ramp"], g["smap"], g["per_timeslot"], g["L"], g["taps"], g["window"],
               g["shifts"], list(g["preambles"]))


@pytest.mark.parametrize("name", tx_golden_names())
def test_transmitter_matches_pygfdm_frames(name):
    """python/qa_transmitter_cc.py:80-183 (single port and cyclic-delay diversity, 5 places) on the fused kernel."""
    import gfdm_amd
    g = load_tx_golden(name)
    tx = _tx_from_golden(g, gfdm_amd.Transmitter)
    nsym = g["symbols"].shape[1]
    assert tx.input_vector_size() == g["A"] * g["M"] and tx.output_vector_size() == g["pygfdm_frames"].shape[-1]
    assert tx.cyclic_shifts() == [int(s) for s in g["shifts"]]
    frames = tx.transmit(g["symbols"], ninput_size=nsym)
    assert len(frames) == len(g["shifts"])
    for port, fr in enumerate(frames):
        assert rel_err(fr, g["pygfdm_frames"][port]) < TOL
        assert_places(fr, g["pygfdm_frames"][port], 5)
    assert rel_err(tx.generic_work(g["symbols"], nsym), g["pygfdm_frames"][0]) < TOL
    blocks = tx.modulate(g["symbols"], nsym)                    # transmitter_kernel::modulate
    assert rel_err(blocks, g["pygfdm_blocks"]) < TOL
    for port, s in enumerate(g["shifts"]):                      # transmitter_kernel::add_frame
        assert rel_err(tx.add_frame(blocks, int(s)), g["pygfdm_frames"][port]) < TOL


def test_transmitter_device_path_and_oracles_at_batch():
    import torch
    import gfdm_amd
    from gfdm_amd import synth
    g = load_tx_golden("tx_ref_k64_m9_cdd")
    tx = _tx_from_golden(g, gfdm_amd.Transmitter)
    assert tx.kernel_name() == "rowlane"
    B, n = 513, tx.input_vector_size()
    sym = synth.qpsk_symbols(0, B, n, torch.device("cuda:0"))
    frames = tx.transmit(sym)
    blocks = tx.modulate(sym)
    torch.cuda.synchronize()
    co = c_oracle.COracleTx(g["M"], g["K"], g["A"], g["cp"], g["cs"], g["ramp"], g["smap"], g["per_timeslot"], g["L"], g["taps"], g["window"],
                            g["shifts"], g["preambles"])
    sym_h = sym.cpu().numpy()
    nt = R.normalize_taps(g["taps"], g["M"])
    assert rel_err(blocks[:9].cpu().numpy(), R.modulate(R.map_to_resources(sym_h[:9], g["M"], g["K"], g["smap"], True), nt, g["M"], g["K"], g["L"])) < TOL
    for port, s in enumerate(g["shifts"]):
        got = frames[port].cpu().numpy()
        assert rel_err(got, co.work(sym_h, port)) < TOL
        ref = R.transmit(sym_h[:9], nt, g["M"], g["K"], g["L"], g["smap"], g["per_timeslot"], g["cp"], g["cs"], g["ramp"], g["window"], int(s),
                         g["preambles"][port])
        assert rel_err(got[:9], ref) < TOL
        assert rel_err(tx.add_frame(blocks, int(s)).cpu().numpy(), got) < 1e-6       # two kernels, same arithmetic up to instruction scheduling


def test_transmitter_generic_family_and_validation():
    import gfdm_amd
    rng = np.random.default_rng(5)
    M, K, A, L = 7, 12, 8, 2                                   # not a compiled shape, K not a power of two
    taps = get_frequency_domain_filter("rrc", 0.3, M, K, L)
    smap = np.array([1, 2, 3, 4, 7, 8, 10, 11])
    cp, cs, ramp = 5, 3, 2
    window = np.concatenate((np.linspace(0.1, 0.9, ramp), np.ones(M * K + cp + cs - 2 * ramp), np.linspace(0.9, 0.1, ramp))).astype(complex)
    pre = [rng.standard_normal(11) + 1j * rng.standard_normal(11) for _ in range(2)]
    nt = R.normalize_taps(taps, M)
    for per_ts in (True, False):
        with gfdm_amd.generic_family_for_testing():
            tx = gfdm_amd.Transmitter(M, K, A, cp, cs, ramp, smap[::-1], per_ts, L, taps, window, [0, 2], pre)   # unsorted map: the reference sorts it
        txj = gfdm_amd.Transmitter(M, K, A, cp, cs, ramp, smap[::-1], per_ts, L, taps, window, [0, 2], pre)     # K = 12: one radix-12 pass, run-time instantiated
        assert (tx.kernel_name(), txj.kernel_name()) == ("generic_lds", "rowlane_jit")
        sym = qpsk(rng, (4, A * M - 3))                        # fewer symbols than slots: the rest is zero
        for port, s in enumerate((0, 2)):
            ref = R.transmit(sym, nt, M, K, L, smap, per_ts, cp, cs, ramp, window, s, pre[port])
            assert rel_err(tx.transmit(sym, ninput_size=A * M - 3)[port], ref) < TOL
            assert rel_err(txj.transmit(sym, ninput_size=A * M - 3)[port], ref) < TOL
        # only the 2*ramp_len window taps given (lib/add_cyclic_prefix_cc.cc:42-56)
        short_window = np.concatenate((window[:ramp], window[-ramp:]))
        tx2 = gfdm_amd.Transmitter(M, K, A, cp, cs, ramp, smap, per_ts, L, taps, short_window, [0, 2], pre)
        assert np.array_equal(tx2.transmit(sym, ninput_size=A * M - 3)[1], txj.transmit(sym, ninput_size=A * M - 3)[1])
    with pytest.raises(ValueError, match="MUST be unique"):
        gfdm_amd.Transmitter(M, K, A, cp, cs, ramp, [1, 1, 3, 4, 7, 8, 10, 11], True, L, taps, window, [0], pre[:1])
    with pytest.raises(ValueError, match="number of window taps"):
        gfdm_amd.Transmitter(M, K, A, cp, cs, ramp, smap, True, L, taps, window[:-1], [0], pre[:1])
    with pytest.raises(ValueError, match="do not match"):
        gfdm_amd.Transmitter(M, K, A, cp, cs, ramp, smap, True, L, taps, window, [0, 1], pre[:1])
    with pytest.raises(ValueError, match="MUST be equal to active_subcarriers"):
        gfdm_amd.Transmitter(M, K, A + 1, cp, cs, ramp, smap, True, L, taps, window, [0], pre[:1])
    tx = gfdm_amd.Transmitter(M, K, A, cp, cs, ramp, smap, True, L, taps, window, [0], pre[:1])
    with pytest.raises(ValueError, match="MUST not exceed"):
        tx.transmit(np.zeros(A * M + 1, np.complex64), ninput_size=A * M + 1)
    with pytest.raises(ValueError, match="no preamble"):
        tx.add_frame(np.zeros(M * K, np.complex64), 1)


def test_transmitter_pybind_surface():
    """gfdm_python.Transmitter = gr::gfdm::transmitter_kernel (C++ class over the C-ABI): frames of every cyclic shift."""
    import gfdm_python
    g = load_tx_golden("tx_ref_k64_m9_cdd")
    tx = gfdm_python.Transmitter(g["M"], g["K"], g["A"], g["cp"], g["cs"], g["ramp"], g["smap"].tolist(), g["per_timeslot"], g["L"],
                                 g["taps"], g["window"], g["shifts"].tolist(), [p for p in g["preambles"]])
    assert tx.input_vector_size() == g["A"] * g["M"] and tx.output_vector_size() == g["pygfdm_frames"].shape[-1]
    assert list(tx.cyclic_shifts()) == [int(s) for s in g["shifts"]]
    frames = tx.transmit(g["symbols"])
    assert len(frames) == len(g["shifts"])
    for port, fr in enumerate(frames):
        assert fr.dtype == np.complex64 and fr.shape == g["pygfdm_frames"][port].shape
        assert_places(fr, g["pygfdm_frames"][port], 5)
    with pytest.raises(ValueError, match="do not match"):
        gfdm_python.Transmitter(g["M"], g["K"], g["A"], g["cp"], g["cs"], g["ramp"], g["smap"].tolist(), True, g["L"], g["taps"], g["window"],
                                [0, 1], [g["preambles"][0]])
