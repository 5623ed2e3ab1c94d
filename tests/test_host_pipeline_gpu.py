"""The host-buffer batch path (gr-gfdm_amd/csrc/gfdm_hostpipe.{h,hip}): what the reference's GNU Radio wrappers call -- HOST pointers, a run
of blocks per scheduler call, three pointers advanced per block
    lib/simple_modulator_cc_impl.cc:62-80, lib/simple_receiver_cc_impl.cc:61-77, lib/advanced_receiver_sb_cc_impl.cc:86-123,
    lib/transmitter_cc_impl.cc:165-177.
The route a call takes (in place on registered memory, one chunk, a chunked bounce through pinned staging sets, kernels across the link or copy
engines, with or without the copy threads) must not change a single bit of the result: every case is compared for EQUALITY with the
device-pointer path on the same blocks, for ragged block counts around the chunk size (0, 1, chunk - 1, chunk, chunk + 1, many chunks + 1) and
at 65 537 blocks with the automatic chunk plan."""
import numpy as np
import pytest

import gfdm_ref as R
from conftest import have_gpu, load_est_golden
from gfdm_amd.filters import get_frequency_domain_filter

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    if not have_gpu():
        pytest.fail("no MI355X visible: the HIP path cannot run (there is no CPU fallback to test instead)")


@pytest.fixture()
def pipeline():
    """restores the process-wide settings of the host path after a test that changes them"""
    import gfdm_amd
    prev = gfdm_amd.get_host_pipeline()
    yield gfdm_amd
    gfdm_amd.set_host_pipeline(*prev)


def qpsk(rng, shape):
    return (((1 - 2 * rng.integers(0, 2, shape)) + 1j * (1 - 2 * rng.integers(0, 2, shape))) / np.sqrt(2)).astype(np.complex64)


def device_reference(call, *arrays):
    """the same entry point on device-resident tensors (torch holds the memory), back on the host"""
    import torch
    dev = [None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrays]
    out = call(*dev)
    torch.cuda.synchronize()
    return out.cpu().numpy()


CHUNK = 8      # blocks per chunk in the ragged tests


@pytest.mark.parametrize("mode,threads,streams", [(0, 0, 1), (0, 2, 2), (0, 0, 2), (1, 0, 1), (1, 2, 2), (2, 2, 1), (2, 0, 2), (3, 1, 1)])
def test_chunked_bounce_equals_the_device_path_for_ragged_block_counts(pipeline, mode, threads, streams):
    g = pipeline
    g.set_host_pipeline(kernel_streams=streams)
    M, K, L = 9, 64, 2
    N = M * K
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    mod, dem = g.Modulator(M, K, L, taps), g.Demodulator(M, K, L, taps)
    adv = g.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points())
    rng = np.random.default_rng(4)
    nmax = 5 * CHUNK + 1
    sym = qpsk(rng, (nmax, N))
    frames = device_reference(lambda s: mod.modulate(s), sym)
    feq = (1.0 + 0.3 * (rng.standard_normal((nmax, N)) + 1j * rng.standard_normal((nmax, N)))).astype(np.complex64)
    ref_mod, ref_mf = frames, device_reference(lambda x: dem.demodulate(x), frames)
    ref_zf = device_reference(lambda x, e: dem.demodulate_equalize(x, e), frames, feq)
    ref_ic = device_reference(lambda x, e: adv.demodulate_equalize(x, e), frames, feq)
    # chunk size in bytes of ALL staged operands: MF / modulate stage 2 x 8 N bytes per block, ZF 3 x 8 N
    for nb in (0, 1, CHUNK - 1, CHUNK, CHUNK + 1, 3 * CHUNK, nmax):
        g.set_host_pipeline(mode, CHUNK * 16 * N, 3, threads)
        out = mod.modulate(sym[:nb])
        assert out.shape == (nb, N) and np.array_equal(out, ref_mod[:nb])
        st = g.host_call_stats()
        if nb:
            assert st["chunks"] == -(-nb // CHUNK) and st["chunk_blocks"] == min(CHUNK, nb) and st["direct_mask"] == 0 and st["mode"] == mode
            assert st["staged_bytes"] == 16 * N * nb
        assert np.array_equal(dem.demodulate(frames[:nb]), ref_mf[:nb])
        g.set_host_pipeline(mode, CHUNK * 24 * N, 2, threads)
        assert np.array_equal(dem.demodulate_equalize(frames[:nb], feq[:nb]), ref_zf[:nb])
        assert np.array_equal(adv.demodulate_equalize(frames[:nb], feq[:nb]), ref_ic[:nb])
        if nb:
            assert g.host_call_stats()["chunks"] == -(-nb // CHUNK)
    # one block per chunk, a single staging set: the degenerate pipeline
    g.set_host_pipeline(mode, 1, 1, threads)
    assert np.array_equal(adv.demodulate_equalize(frames[:5], feq[:5]), ref_ic[:5])
    assert g.host_call_stats()["chunks"] == 5


def test_registered_buffers_are_used_in_place(pipeline):
    g = pipeline
    M, K, L = 9, 64, 2
    N = M * K
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    mod = g.Modulator(M, K, L, taps)
    adv = g.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points())
    rng = np.random.default_rng(5)
    nb = 300
    sym = qpsk(rng, (nb, N))
    # registration takes whole pages the caller owns: page-aligned arrays (gfdm_amd.aligned_empty / aligned_copy)
    frames = g.aligned_copy(mod.modulate(sym))
    feq = g.aligned_copy((1.0 + 0.3 * (rng.standard_normal((nb, N)) + 1j * rng.standard_normal((nb, N)))).astype(np.complex64))
    ref = adv.demodulate_equalize(frames, feq)                        # bounced (not registered yet)
    assert g.host_call_stats()["direct_mask"] == 0 and g.host_call_stats()["chunks"] >= 2
    out = g.aligned_empty(ref.shape)
    with pytest.raises(ValueError):
        g.register_host(np.empty(10000, np.complex64)[3:])           # not on a page boundary: refused (a page shared with other objects must not be pinned)
    assert g.lib().gfdm_hip_register_host(frames.ctypes.data, 1000) == g.capi.EINVAL      # ... and so is a size that is not whole pages
    with g.registered_host(out, frames, feq):
        res = adv.demodulate_equalize(frames, feq, out=out)
        st = g.host_call_stats()
        assert res is out and st["direct_mask"] == 0b111 and st["staged_bytes"] == 0
        assert st["chunks"] == 1                      # kernels on the caller's memory: one launch
        assert np.array_equal(out, ref)
        for mode in (1, 2, 3):                        # copy engines between the caller's memory and device staging, chunk by chunk
            g.set_host_pipeline(mode, 7 * 24 * N, 3, 0)
            out[:] = 0
            adv.demodulate_equalize(frames, feq, out=out)
            st = g.host_call_stats()
            assert st["chunks"] == -(-nb // 7) and st["direct_mask"] == 0b111 and st["staged_bytes"] == 0 and np.array_equal(out, ref)
        g.set_host_pipeline(0, 0, 3, 2)
        # any part of a registered buffer: interior pointers, ragged counts
        out[:] = 0
        adv.demodulate_equalize(frames[7:130], feq[7:130], out=out[7:130])
        assert g.host_call_stats()["direct_mask"] == 0b111
        assert np.array_equal(out[7:130], ref[7:130]) and not out[:7].any() and not out[130:].any()
        # in place (out is in): the reference's generic_work copies its input first, so callers may rely on it -- the output is bounced
        buf = g.aligned_copy(frames)
        g.register_host(buf)
        try:
            mf = g.Demodulator(M, K, L, taps)
            want = mf.demodulate(frames)
            mf.demodulate(buf, out=buf)
            assert g.host_call_stats()["direct_mask"] == 0b10          # operand 0 (out) staged, operand 1 (in) in place
            assert np.array_equal(buf, want)
        finally:
            g.unregister_host(buf)
    # mixed: only the input registered
    with g.registered_host(frames):
        g.set_host_pipeline(0, CHUNK * 16 * N, 3, 0)
        assert np.array_equal(adv.demodulate_equalize(frames, feq), ref)
        st = g.host_call_stats()
        assert st["direct_mask"] == 0b010 and st["chunks"] == -(-nb // CHUNK) and st["staged_bytes"] == 16 * N * nb
    # after unregistering the memory is ordinary again
    adv.demodulate_equalize(frames, feq)
    assert g.host_call_stats()["direct_mask"] == 0
    with pytest.raises(Exception):
        g.unregister_host(frames)


def test_65537_blocks_with_the_automatic_chunk_plan(pipeline):
    g = pipeline
    M, K, L = 9, 64, 2
    N = M * K
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    mod, dem = g.Modulator(M, K, L, taps), g.Demodulator(M, K, L, taps)
    nb = 65537
    rng = np.random.default_rng(6)
    sym = qpsk(rng, (nb, N))
    frames = g.aligned_empty((nb, N))
    mod.modulate(sym, out=frames)
    st = g.host_call_stats()
    assert st["chunks"] > 8 and st["direct_mask"] == 0 and st["copy_threads"] >= 1
    assert np.array_equal(frames, device_reference(lambda s: mod.modulate(s), sym))
    out = dem.demodulate(frames)
    assert np.array_equal(out, device_reference(lambda x: dem.demodulate(x), frames))
    # size-independent property at the full size: the matched-filter receiver returns the symbols up to the filter's self-interference
    assert np.max(np.abs(out[-1] - sym[-1])) < 0.6 and np.array_equal(np.sign(out.real[::4097]), np.sign(sym.real[::4097]))
    reg = g.aligned_empty(out.shape)
    with g.registered_host(frames, reg):
        dem.demodulate(frames, out=reg)
        assert g.host_call_stats()["chunks"] == 1 and g.host_call_stats()["staged_bytes"] == 0
    assert np.array_equal(reg, out)


def test_other_entry_points_through_the_chunked_bounce(pipeline):
    """frames + demapper, the self-estimating receiver with preambles at a stride inside a burst buffer, the stand-alone estimator with its
    float outputs (estimate_snr), the composite transmitter with two output ports, mapper and prefixer: chunked == one chunk"""
    g = pipeline
    e = load_est_golden("est_cfg2_m9_k64_a52")
    M, K, A = e["M"], e["K"], e["A"]
    L, N = 2, M * K
    rng = np.random.default_rng(7)
    smap = np.concatenate((np.arange(1, A // 2 + 1), np.arange(K - A // 2, K)))
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    est = g.ChannelEstimator(M, K, A, True, 1, e["preamble"])
    adv = g.AdvancedReceiver(M, K, L, taps, smap, 2, R.qpsk_points())
    cp, nb = 16, 37
    adv.configure_frames(cp + N + 8, cp, smap, True)
    adv.set_channel_estimator(est)
    burst_len = 2 * K + cp + N + 8
    bursts = (rng.standard_normal((nb, burst_len)) + 1j * rng.standard_normal((nb, burst_len))).astype(np.complex64)
    mapper = g.ResourceMapper(M, K, A, smap, True)
    prefixer = g.CyclicPrefixer(N, cp, 8, 4, np.ones(2 * 4, np.complex64), 0)
    tx = g.Transmitter(M, K, A, cp, 8, 4, smap, True, L, taps, np.ones(2 * 4, np.complex64), [0, 3],
                       (rng.standard_normal((2, 2 * K)) + 1j * rng.standard_normal((2, 2 * K))).astype(np.complex64))
    data = qpsk(rng, (nb, A * M))
    blocks = qpsk(rng, (nb, N))

    def everything():
        frames = np.ascontiguousarray(bursts[:, 2 * K:])
        res = [adv.demodulate_frames(frames, None)]
        res.append(adv.demodulate_estimated(frames, np.ascontiguousarray(bursts[:, :2 * K])))                  # packed preambles
        res.append(adv.demodulate_estimated(frames, bursts.reshape(-1), preamble_stride=burst_len))           # preamble b at b * burst_len
        assert np.array_equal(res[-1], res[-2])
        res.append(est.estimate_frame(np.ascontiguousarray(bursts[:, :2 * K])))
        res.extend(est.estimate_snr(np.ascontiguousarray(bursts[:, :2 * K])))
        res.extend(tx.generic_work(data))
        res.append(mapper.map_to_resources(data))
        res.append(mapper.demap_from_resources(blocks))
        res.append(prefixer.add_cyclic_prefix(blocks))
        return [np.asarray(r) for r in res]

    g.set_host_pipeline(0, 1 << 30, 3, 0)
    whole = everything()
    for mode, chunk, depth, threads in ((0, 5 * 8 * N, 3, 2), (1, 3 * 8 * N, 2, 0), (0, 1, 4, 0), (2, 4 * 8 * N, 3, 1), (3, 9 * 8 * N, 2, 0)):
        g.set_host_pipeline(mode, chunk, depth, threads)
        parts = everything()
        assert len(parts) == len(whole)
        for a, b in zip(parts, whole):
            assert a.shape == b.shape and np.array_equal(a, b)


def test_copy_threads_are_stopped_by_quiesce_and_come_back(pipeline):
    g = pipeline
    M, K, L = 9, 64, 2
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    mod = g.Modulator(M, K, L, taps)
    sym = qpsk(np.random.default_rng(8), (2048, M * K))
    g.set_host_pipeline(0, 0, 3, 3)
    a = mod.modulate(sym)
    assert g.host_call_stats()["copy_threads"] >= 1
    g.quiesce()
    b = mod.modulate(sym)
    assert g.host_call_stats()["copy_threads"] >= 1 and np.array_equal(a, b)
    with pytest.raises(ValueError):
        g.set_host_pipeline(5, -1, -1, -1)


def test_sharded_host_batches_run_their_devices_concurrently(pipeline):
    """gfdm_amd.sharding.ShardedBatch.run_global (one process driving several GPUs with HOST batches): one host thread per local device, as
    gr::gfdm::sharded_batch does in C++.  Device 0 listed four times: the results are those of one handle, the wall time is clearly below
    the sum of the four host calls made one after the other, and a shard's exception reaches the caller after all threads have joined."""
    import time
    from gfdm_amd import sharding
    g = pipeline
    M, K, L = 9, 64, 2
    N, total = M * K, 4 * 4096
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    sb = sharding.ShardedBatch(lambda d: g.Demodulator(M, K, L, taps, device=d), [0, 0, 0, 0])
    x = qpsk(np.random.default_rng(9), (total, N))
    g.set_host_pipeline(copy_threads=0)          # every shard's thread does its own bounce copies: the test measures the threads of run_global
    whole = g.Demodulator(M, K, L, taps).demodulate(x)
    parts = sb.run_global("demodulate", [x], [N])
    assert [(s, n) for s, n, _ in parts] == [(i * 4096, 4096) for i in range(4)]
    # the statistics of a host call belong to the thread that made it: run_global keeps every shard's own
    assert len(sb.last_host_call_stats) == 4 and all(st["chunks"] >= 2 and st["staged_bytes"] == 16 * N * 4096 for st in sb.last_host_call_stats)
    assert np.array_equal(np.concatenate([p for _, _, p in parts]), whole)
    serial, threaded = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        for i, k in enumerate(sb.kernels):
            k.demodulate(x[i * 4096:(i + 1) * 4096])
        serial.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        sb.run_global("demodulate", [x], [N])
        threaded.append(time.perf_counter() - t0)
    print("run_global: four host shards of 4096 blocks one after the other %.2f ms, on four threads %.2f ms" % (min(serial) * 1e3, min(threaded) * 1e3))
    assert min(threaded) < 0.9 * min(serial)          # measured 3.4 against 5.0 ms; the bound leaves room for a busy box

    class Boom(RuntimeError):
        pass

    def bad(*a):
        raise Boom("shard failed")
    sb.kernels[2].demodulate = bad
    with pytest.raises(Boom):
        sb.run_global("demodulate", [x], [N])


@pytest.mark.parametrize("M,K,L,alpha,nb", [(15, 128, 4, 0.2, 300), (31, 256, 2, 0.1, 70), (5, 32, 2, 0.5, 1000), (21, 37, 2, 0.35, 40), (7, 12, 2, 0.3, 500)])
def test_baseline_shapes_and_other_kernel_families_through_the_host_path(pipeline, M, K, L, alpha, nb):
    """BASELINE configs[3] (multi-wavefront blocks, IC rounds on the matrix cores), configs[4] (62 KB tiles), configs[0], a shape of the generic family
    and a run-time instantiated one: their kernels reading and writing HOST memory (chunked bounce, and in place on registered buffers) give
    exactly what they give on device memory."""
    g = pipeline
    N = M * K
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    mod, dem = g.Modulator(M, K, L, taps), g.Demodulator(M, K, L, taps)
    adv = g.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points())
    rng = np.random.default_rng(M * K)
    sym = g.aligned_copy(qpsk(rng, (nb, N)))
    feq = g.aligned_copy((1.0 + 0.3 * (rng.standard_normal((nb, N)) + 1j * rng.standard_normal((nb, N)))).astype(np.complex64))
    frames = g.aligned_copy(device_reference(lambda s: mod.modulate(s), sym))
    refs = (frames, device_reference(lambda x: dem.demodulate(x), frames), device_reference(lambda x, e: dem.demodulate_equalize(x, e), frames, feq),
            device_reference(lambda x: adv.demodulate(x), frames), device_reference(lambda x, e: adv.demodulate_equalize(x, e), frames, feq))
    calls = (lambda o: mod.modulate(sym, out=o), lambda o: dem.demodulate(frames, out=o), lambda o: dem.demodulate_equalize(frames, feq, out=o),
             lambda o: adv.demodulate(frames, out=o), lambda o: adv.demodulate_equalize(frames, feq, out=o))
    g.set_host_pipeline(0, 7 * 16 * N, 3, 2, 2)                        # seven blocks per chunk (MF) / four (ZF)
    for call, ref in zip(calls, refs):
        assert np.array_equal(call(None), ref)
    assert g.host_call_stats()["chunks"] > 3
    out = g.aligned_empty((nb, N))
    with g.registered_host(sym, frames, feq, out):
        g.set_host_pipeline(0, 0, 3, 2, 2)
        for call, ref in zip(calls, refs):
            out[:] = 0
            call(out)
            assert g.host_call_stats()["staged_bytes"] == 0 and np.array_equal(out, ref)


def test_a_call_across_two_registrations(pipeline):
    """two buffers registered one after the other, adjacent in the address space, and a call that runs across the seam: in place only if the GPU
    sees them contiguously too, bounced otherwise -- the same result either way, never a fault"""
    g = pipeline
    M, K, L = 9, 64, 2
    N = M * K
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    dem = g.Demodulator(M, K, L, taps)
    nb = 64                                              # 64 blocks x 4608 B = 72 pages: the halves are page aligned
    x = g.aligned_copy(qpsk(np.random.default_rng(11), (nb, N)))
    ref = dem.demodulate(x)
    lo, hi = x[:nb // 2], x[nb // 2:]
    g.register_host(lo)
    g.register_host(hi)
    try:
        out = dem.demodulate(x)
        st = g.host_call_stats()
        assert np.array_equal(out, ref) and st["direct_mask"] in (0b00, 0b10)
        out = dem.demodulate(x[5:nb // 2 - 1])           # inside one registration
        assert np.array_equal(out, ref[5:nb // 2 - 1]) and g.host_call_stats()["direct_mask"] == 0b10
    finally:
        g.unregister_host(lo)
        g.unregister_host(hi)


def test_memory_pinned_or_owned_by_somebody_else(pipeline):
    """hipHostMalloc'ed memory (a torch pinned tensor) and device memory handed to a *_host entry point are used in place as well: the runtime vouches for the
    whole allocation (hipMemGetAddressRange), no registration call is needed."""
    import ctypes
    import torch
    g = pipeline
    M, K, L = 9, 64, 2
    N, nb = M * K, 300
    dem = g.Demodulator(M, K, L, get_frequency_domain_filter("rrc", 0.2, M, K, L))
    xp, op = torch.empty(nb, N, dtype=torch.complex64).pin_memory(), torch.empty(nb, N, dtype=torch.complex64).pin_memory()
    x, o = xp.numpy(), op.numpy()
    x[...] = qpsk(np.random.default_rng(12), (nb, N))
    ref = dem.demodulate(np.array(x))                                   # a pageable copy: bounced
    assert g.host_call_stats()["direct_mask"] == 0
    dem.demodulate(x, out=o)
    st = g.host_call_stats()
    assert st["direct_mask"] == 0b11 and st["staged_bytes"] == 0 and np.array_equal(o, ref)
    xd = xp.cuda()
    od = torch.empty_like(xd)
    assert g.lib().gfdm_hip_receiver_demodulate_host(dem._h, ctypes.c_void_p(od.data_ptr()), ctypes.c_void_p(xd.data_ptr()), None, ctypes.c_int64(nb)) == 0
    assert g.host_call_stats()["direct_mask"] == 0b11 and np.array_equal(od.cpu().numpy(), ref)
    # device memory is never bounced (a CPU copy would dereference a device address): an interior slice of a device allocation is used in place as
    # well, and a device output that overlaps its input is refused instead of being staged
    big = torch.zeros(nb + 10, N, dtype=torch.complex64, device="cuda")
    big[5:5 + nb] = xd
    od.zero_()
    assert g.lib().gfdm_hip_receiver_demodulate_host(dem._h, ctypes.c_void_p(od.data_ptr()), ctypes.c_void_p(big[5].data_ptr()), None, ctypes.c_int64(nb)) == 0
    assert g.host_call_stats()["direct_mask"] == 0b11 and np.array_equal(od.cpu().numpy(), ref)
    assert g.lib().gfdm_hip_receiver_demodulate_host(dem._h, ctypes.c_void_p(big[6].data_ptr()), ctypes.c_void_p(big[5].data_ptr()), None, ctypes.c_int64(nb)) == g.capi.EINVAL
    assert b"overlaps" in g.lib().gfdm_hip_last_error()


@pytest.mark.parametrize("threads", [0, 3])
def test_streaming_store_copies_move_the_same_bytes_from_and_to_unaligned_buffers(pipeline, threads):
    """Calls that stage 2 MiB or more copy with non-temporal stores (gfdm_hostpipe.hip: copy_streaming_avx2, destination aligned to 32 bytes with
    a memcpy'd head and tail).  Buffers that start 8 bytes off any alignment, chunk sizes that leave ragged pieces: equal to the plain-memcpy
    route and to the device path, and nothing written outside the output."""
    g = pipeline
    M, K, L = 9, 64, 2
    N = M * K
    dem = g.Demodulator(M, K, L, get_frequency_domain_filter("rrc", 0.2, M, K, L))
    nb = 1200                                       # 5.5 MB in, 5.5 MB out
    rng = np.random.default_rng(21)
    raw_in = np.zeros(nb * N + 7, np.complex64)
    raw_out = np.full(nb * N + 7, np.complex64(77 - 5j))
    ref = None
    for off in (0, 1, 3):                           # complex64 elements: 0, 8, 24 bytes off the allocation's alignment
        x = raw_in[off:off + nb * N].reshape(nb, N)
        x[...] = qpsk(rng, (nb, N))
        want = device_reference(dem.demodulate, x)
        for chunk in (0, 999_983, 3 << 20):
            for streaming in (1, 0):
                g.set_host_pipeline(0, chunk, 3, threads, 2)
                prev = g.lib().gfdm_hip_set_host_streaming_copies_for_testing(streaming)
                try:
                    raw_out[...] = np.complex64(77 - 5j)
                    y = raw_out[off:off + nb * N].reshape(nb, N)
                    dem.demodulate(x, out=y)
                finally:
                    g.lib().gfdm_hip_set_host_streaming_copies_for_testing(prev)
                assert np.array_equal(y, want), (off, chunk, streaming)
                assert np.all(raw_out[:off] == np.complex64(77 - 5j)) and np.all(raw_out[off + nb * N:] == np.complex64(77 - 5j))


def test_automatic_chunk_plan_of_small_and_mid_size_calls(pipeline):
    """chunk_bytes = 0: one chunk up to 512 KiB of staged bytes, else round(sqrt(1.3 x MiB staged)) chunks, at least two (gfdm_hostpipe.hip; the measured
    optimum of profiles/r04/host_chunk_sweep.txt).  K=64 M=9 MF demodulation stages 9216 bytes per block."""
    g = pipeline
    g.set_host_pipeline(0, 0, 3, 3, 2)
    M, K, L = 9, 64, 2
    N = M * K
    dem = g.Demodulator(M, K, L, get_frequency_domain_filter("rrc", 0.2, M, K, L))
    x = qpsk(np.random.default_rng(31), (4096, N))
    want = device_reference(dem.demodulate, x)
    for nb, chunks in ((1, 1), (48, 1), (56, 1), (57, 2), (64, 2), (257, 2), (512, 2), (1024, 3), (4096, 7)):
        out = dem.demodulate(x[:nb])
        st = g.host_call_stats()
        assert st["chunks"] == chunks and st["chunk_blocks"] == -(-nb // chunks), (nb, st)
        assert (st["copy_threads"] >= 1) == (st["chunk_blocks"] * 2 * N * 8 >= (1 << 20)), (nb, st)      # the pool takes copy jobs of 1 MiB and more
        assert np.array_equal(out, want[:nb]), nb


def test_a_stale_hip_error_of_the_application_does_not_fail_the_next_call():
    """hipGetLastError() is sticky: it keeps the error of any earlier failed runtime call of the thread until somebody reads it, and the launchers check their
    launches with it.  An application call that failed (here: an allocation nobody could satisfy) must not make the next GFDM call fail -- found by the injected
    failures of tests/sanitize, fixed at call entry (DeviceGuard) and in front of the completion-ticket launch; this is the same on the real runtime."""
    import ctypes
    import torch
    import gfdm_amd
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    M, K, L = 9, 64, 2
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    mod, dem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
    rng = np.random.default_rng(3)
    d = qpsk(rng, (5, M * K))
    want = mod.modulate(d)
    want_rx = dem.demodulate(want)
    for entry in ("host", "device"):
        p = ctypes.c_void_p()
        if entry == "host":
            assert hip.hipMalloc(ctypes.byref(p), 1 << 60) != 0      # fails, and leaves its error behind in this thread
            got = mod.modulate(d)                                     # numpy in: the *_host entry point (launch + completion ticket)
            got_rx = dem.demodulate(got)
        else:
            dt = torch.from_numpy(d).cuda()                           # (torch first: its own launch checks trip over a stale error just the same)
            o1, o2 = torch.empty_like(dt), torch.empty_like(dt)
            torch.cuda.synchronize()
            assert hip.hipMalloc(ctypes.byref(p), 1 << 60) != 0      # no torch call between the failure and ours
            mod.modulate(dt, out=o1)                                  # tensors in: the *_device entry point
            assert hip.hipMalloc(ctypes.byref(p), 1 << 60) != 0
            dem.demodulate(o1, out=o2)
            torch.cuda.synchronize()
            got, got_rx = o1.cpu().numpy(), o2.cpu().numpy()
        assert np.array_equal(got, want) and np.array_equal(got_rx, want_rx)
