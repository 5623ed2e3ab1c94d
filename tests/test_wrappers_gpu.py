"""SURVEY.md section 8(f) row 4 -- the GNU Radio wrappers' work() bodies with ONE batched launch per scheduler call
(gr-gfdm_amd/cpp/include/gfdm/batched_work.h), driven by a scheduler stand-in (the test-only module gfdm_testing) that calls them the way
GNU Radio calls work(): successive, ragged runs of noutput_items, pointers advanced by what work() returned.  Checked against the
oracle on the whole stream and against the item accounting of the reference loops:
    lib/simple_modulator_cc_impl.cc:62-80, lib/simple_receiver_cc_impl.cc:61-77   process floor(n / block_size) blocks, return n
    lib/advanced_receiver_sb_cc_impl.cc:86-123                                     return n_blocks * block_size; port 1 optional
    lib/transmitter_cc_impl.cc:130-195                                             frames = min(nout / out_size, nin / in_size)
    lib/channel_estimator_cc_impl.cc:88-120                                        frames = nout / frame_len, snr + cnr per frame
    lib/resource_mapper_cc_impl.cc:86-106, lib/resource_demapper_cc_impl.cc:87-105 frames = min(nout / out_size, nin / in_size)
    lib/cyclic_prefixer_cc_impl.cc:93-110                                          frames = nout / frame_size
Also here: the reference's legacy 2-D receiver API (lib/receiver_kernel_cc.cc:130-163,194-209,227-272) and
gfdm_kernel_utils::calculate_signal_energy (lib/gfdm_kernel_utils.cc:59-65), which no other test reaches."""
import numpy as np
import pytest

import gfdm_ref as R
from conftest import have_gpu, load_est_golden, load_tx_golden, rel_err
from gfdm_amd.filters import get_frequency_domain_filter

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    if not have_gpu():
        pytest.fail("no MI355X visible: the HIP path cannot run (there is no CPU fallback to test instead)")


def qpsk(rng, shape):
    return ((1 - 2 * rng.integers(0, 2, shape)) + 1j * (1 - 2 * rng.integers(0, 2, shape))) / np.sqrt(2)


@pytest.mark.parametrize("M,K,L,alpha", [(9, 64, 2, 0.2), (5, 32, 2, 0.5), (21, 12, 2, 0.35)])
def test_sync_block_work_bodies_with_ragged_scheduler_calls(M, K, L, alpha):
    import gfdm_python
    import gfdm_testing as T
    rng = np.random.default_rng(M * K)
    bs = M * K
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    nt = R.normalize_taps(taps, M)
    chunks = [3 * bs, 0, bs, 7 * bs, 2 * bs, 19 * bs]                 # what a scheduler hands a block with set_output_multiple(bs)
    nblk = sum(chunks) // bs
    sym = qpsk(rng, nblk * bs)
    mod, dem = gfdm_python.Modulator(M, K, L, taps), gfdm_python.Demodulator(M, K, L, taps)
    frames, ret = T.scheduler_run(mod, sym, chunks)
    assert ret == chunks                                              # work() returns noutput_items
    ref_frames = R.modulate(sym.reshape(nblk, bs), nt, M, K, L)
    assert rel_err(frames.reshape(nblk, bs), ref_frames) < TOL
    out, ret = T.scheduler_run(dem, frames, chunks[::-1])
    assert ret == chunks[::-1]
    assert rel_err(out.reshape(nblk, bs), R.demodulate(ref_frames, nt, M, K, L)) < TOL
    # identical to the per-block loop of the reference wrapper (same kernels, same inputs)
    per_block = np.stack([dem.demodulate(frames[b * bs:(b + 1) * bs]) for b in range(4)])
    assert np.array_equal(per_block.reshape(-1), out[:4 * bs])
    # the scheduler's buffers registered once (gr::gfdm::host_registration, gfdm/host_memory.h): the same work() calls run in place on them
    import gfdm_amd
    out_reg, ret_reg = T.scheduler_run(dem, frames, chunks[::-1], registered=True)
    assert ret_reg == ret and np.array_equal(out_reg, out)
    st = gfdm_amd.host_call_stats()
    assert st["direct_mask"] == 0b11 and st["staged_bytes"] == 0 and st["chunks"] == 1
    T.scheduler_run(dem, frames, chunks[::-1])
    assert gfdm_amd.host_call_stats()["direct_mask"] == 0                       # ... and ordinary memory is bounced again afterwards
    # a run that is not a multiple of the block size: the reference processes the whole blocks and still returns noutput_items
    out2, ret2 = T.scheduler_run(dem, frames[:2 * bs + 7], [2 * bs + 7])
    assert ret2 == [2 * bs + 7] and np.array_equal(out2[:2 * bs], out[:2 * bs]) and not out2[2 * bs:].any()


@pytest.mark.parametrize("with_eq", [False, True])
def test_advanced_receiver_work_body(with_eq):
    import gfdm_python
    import gfdm_testing as T
    M, K, L = 9, 64, 2
    rng = np.random.default_rng(5)
    bs = M * K
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    nt = R.normalize_taps(taps, M)
    smap = np.concatenate((np.arange(1, 27), np.arange(38, 64)))
    chunks = [2 * bs, 5 * bs, 0, bs, 11 * bs + 5]                     # the last run has 5 stray items: returned floor, not n
    nblk = sum(c // bs for c in chunks)
    d = np.zeros((nblk, K, M), complex)
    d[:, smap, :] = qpsk(rng, (nblk, len(smap), M))
    x = R.modulate(d.reshape(nblk, bs), nt, M, K, L)
    feq = np.fft.fft(np.array([1, .5, .1j, .1 + .05j]), bs)[None, :] * np.exp(0.01j * np.arange(nblk))[:, None]
    stream = (np.fft.ifft(np.fft.fft(x, axis=-1) * feq, axis=-1) if with_eq else x).reshape(-1)
    stream = np.concatenate((stream, np.zeros(5)))
    eq_stream = np.concatenate((feq.reshape(-1), np.ones(5))) if with_eq else None
    adv = gfdm_python.AdvancedReceiver(M, K, L, taps, smap.tolist(), 2, gfdm_python.Constellation.qpsk(), 0)
    out, ret = T.scheduler_run_equalize(adv, stream, eq_stream, chunks)
    assert ret == [(c // bs) * bs for c in chunks]                   # lib/advanced_receiver_sb_cc_impl.cc:122
    ref = R.advanced_receive(stream[:nblk * bs].reshape(nblk, bs), nt, M, K, L, smap, R.qpsk_points(), 2, f_eq=feq if with_eq else None, kind="qpsk")
    assert rel_err(out[:nblk * bs].reshape(nblk, bs), ref) < TOL


def test_transmitter_general_work_body():
    import gfdm_python
    import gfdm_testing as T
    g = load_tx_golden("tx_ref_k64_m9_cdd")
    tx = gfdm_python.Transmitter(g["M"], g["K"], g["A"], g["cp"], g["cs"], g["ramp"], g["smap"].tolist(), g["per_timeslot"], g["L"], g["taps"], g["window"],
                                 [int(s) for s in g["shifts"]], [p for p in g["preambles"]])
    nin, nout, ports = tx.input_vector_size(), tx.output_vector_size(), len(g["shifts"])
    rng = np.random.default_rng(2)
    nfr = 12
    sym = qpsk(rng, (nfr, nin))
    # (noutput_items, ninput_items[0]) per general_work call: output-limited, input-limited, nothing to do, the rest
    calls = [(3 * nout + 11, 9 * nin), (8 * nout, 2 * nin + 3), (nout - 1, 5 * nin), (100 * nout, 7 * nin)]
    outs, frames = T.scheduler_run_transmitter(tx, sym.reshape(-1), calls, ports)
    assert frames == [3, 2, 0, 7]
    nt = R.normalize_taps(g["taps"], g["M"])
    for port, s in enumerate(g["shifts"]):
        ref = R.transmit(sym, nt, g["M"], g["K"], g["L"], g["smap"], g["per_timeslot"], g["cp"], g["cs"], g["ramp"], g["window"], int(s),
                         g["preambles"][port])
        assert rel_err(outs[port], ref) < TOL


def test_mapper_demapper_and_prefixer_general_work_bodies():
    import gfdm_python
    import gfdm_testing as T
    g = load_tx_golden("tx_ref_k64_m9_cdd")
    M, K, A = g["M"], g["K"], g["A"]
    rng = np.random.default_rng(4)
    nfr = 12
    for per_ts in (True, False):
        mapper = gfdm_python.Resource_mapper(M, K, A, g["smap"].tolist(), per_ts)
        demapper = gfdm_python.Resource_mapper(M, K, A, g["smap"].tolist(), per_ts, False)
        assert (mapper.input_vector_size(), mapper.output_vector_size()) == (A * M, K * M)
        assert (demapper.input_vector_size(), demapper.output_vector_size()) == (K * M, A * M)
        sym = qpsk(rng, (nfr, A * M)).astype(np.complex64)
        nin, nout = A * M, K * M
        calls = [(3 * nout + 11, 9 * nin), (8 * nout, 2 * nin + 3), (nout - 1, 5 * nin), (100 * nout, 7 * nin)]
        grid, frames = T.scheduler_run_mapper(mapper, True, sym.reshape(-1), calls)
        assert frames == [3, 2, 0, 7]
        assert np.array_equal(grid, R.map_to_resources(sym, M, K, g["smap"], per_ts).astype(np.complex64))
        back, frames = T.scheduler_run_mapper(demapper, False, grid.reshape(-1), [(5 * nin, 4 * nout + 1), (0, 3 * nout), (9 * nin + 2, 8 * nout)])
        assert frames == [4, 0, 8] and np.array_equal(back, sym)
    N = M * K
    pre = gfdm_python.Cyclic_prefixer(N, g["cp"], g["cs"], g["ramp"], list(g["window"]), int(g["shifts"][1]))
    blocks = (rng.standard_normal((nfr, N)) + 1j * rng.standard_normal((nfr, N))).astype(np.complex64)
    F = pre.frame_size()
    out, frames = T.scheduler_run_prefixer(pre, blocks.reshape(-1), [2 * F + 5, F - 1, 0, 7 * F, 3 * F])
    assert frames == [2, 0, 0, 7, 3]
    assert rel_err(out, R.add_cyclic_prefix(blocks, g["cp"], g["cs"], g["ramp"], g["window"], int(g["shifts"][1]))) < 1e-6


def test_channel_estimator_general_work_body():
    import gfdm_python
    import gfdm_testing as T
    g = load_est_golden("est_cfg2_m9_k64_a52")
    M, K, A = g["M"], g["K"], g["A"]
    est = gfdm_python.Preamble_channel_estimator(M, K, A, True, 1, g["preamble"])
    rx = np.concatenate([g["rx_preambles"]] * 3)                      # 12 received preambles
    fl = M * K
    out, frames, tags = T.scheduler_run_estimator(est, rx.reshape(-1), [2 * fl, fl - 1, 5 * fl + 17, 5 * fl])
    assert frames == [2, 0, 5, 5] and out.shape == (12, fl)
    assert rel_err(out, R.estimate_frame(rx, g["preamble"], M, K, A, True)) < TOL
    assert [t[0] for t in tags] == list(range(12))                    # one snr_lin / cnr tag pair per frame, in order
    snr_ref, cnr_ref = R.estimate_snr(rx.astype(np.complex64), K, A, True)
    finite = np.isfinite(snr_ref) & (np.abs(snr_ref) < 1e6)
    assert finite.any()
    for i in np.nonzero(finite)[0]:
        assert abs(tags[i][1] - snr_ref[i]) <= 2e-3 * abs(snr_ref[i])
        assert np.allclose(tags[i][2], cnr_ref[i], rtol=2e-3, atol=1e-3 * np.max(np.abs(cnr_ref[i])))


def test_legacy_2d_receiver_api_and_signal_energy():
    import gfdm_python
    import gfdm_testing as T
    M, K, L = 5, 32, 2                                                # shape of qa_python_bindings.py:388-415
    rng = np.random.default_rng(1)
    taps = get_frequency_domain_filter("rrc", 0.35, M, K, L)
    nt = R.normalize_taps(taps, M)
    dem, mod = gfdm_python.Demodulator(M, K, L, taps), gfdm_python.Modulator(M, K, L, taps)
    d = qpsk(rng, M * K)
    frame = mod.modulate(d)
    S = R.fft_filter_downsample(frame, nt, M, K, L)
    fd2 = T.legacy_filter_superposition(dem, frame)                   # :130-163
    assert fd2.shape == (K, M) and rel_err(fd2.reshape(-1), S) < TOL
    td2 = T.legacy_demodulate_subcarrier(dem, fd2)                    # :194-209
    assert rel_err(td2.reshape(-1), R.transform_subcarriers_to_td(S, M, K)) < TOL
    ic2 = T.legacy_remove_sc_interference(dem, d.reshape(K, M), fd2)  # :244-272 (result replaces sc_symbols)
    assert rel_err(ic2.reshape(-1), R.cancel_sc_interference(d, S, R.ic_filter_taps(nt, M, L), M, K)) < TOL
    mat, back = T.legacy_vectorize_serialize(dem, d)                  # :227-242
    assert np.array_equal(mat, d.reshape(K, M).astype(np.complex64)) and np.array_equal(back, d.astype(np.complex64))
    x = rng.standard_normal(1000) + 1j * rng.standard_normal(1000)
    for k in (dem, mod):                                              # gfdm_kernel_utils.cc:59-65
        assert abs(T.calculate_signal_energy(k, x) - np.sum(np.abs(x.astype(np.complex64)) ** 2)) < 1e-3 * 2000
    assert T.calculate_signal_energy(dem, np.zeros(0, np.complex64)) == 0.0
