"""GPU parity of the stand-alone resource mapper / demapper and cyclic prefixer (gr-gfdm resource_mapper_kernel_cc,
add_cyclic_prefix_cc; Python classes Resource_mapper, Cyclic_prefixer) through the C-ABI (ctypes, host and device entry points) and
through the pybind11 module, against the pygfdm golden vectors and the numpy oracle.

Bar: index work is bit-exact (the kernels only move complex64 values); the two window ramps multiply in float32, compared at 1e-6
relative against the float64 oracle.
"""
import numpy as np
import pytest

import gfdm_ref as R
from conftest import have_gpu, load_tx_golden, rel_err, tx_golden_names

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    if not have_gpu():
        pytest.fail("no MI355X visible: the HIP path cannot run (there is no CPU fallback to test instead)")


def c64(a):
    return np.asarray(a).astype(np.complex64)


def pygfdm_grid(g):
    grid = g["pygfdm_grid"].reshape(g["pygfdm_grid"].shape[0], g["K"], -1)        # ceil(n / A) timeslots in pygfdm, `timeslots` here
    out = np.zeros((grid.shape[0], g["K"], g["M"]), np.complex64)
    out[:, :, :grid.shape[2]] = grid
    return out.reshape(-1, g["K"] * g["M"])


@pytest.mark.parametrize("name", tx_golden_names())
def test_mapper_and_prefixer_match_pygfdm(name):
    import gfdm_amd
    import gfdm_python
    g = load_tx_golden(name)
    M, K, A, plen = g["M"], g["K"], g["A"], g["preambles"].shape[-1]
    nsym = g["symbols"].shape[-1]
    mapper = gfdm_amd.ResourceMapper(M, K, A, g["smap"][::-1], g["per_timeslot"])        # unsorted map: the reference sorts it
    assert (mapper.block_size(), mapper.frame_size()) == (A * M, K * M)
    grid = mapper.map_to_resources(g["symbols"], ninput_size=nsym)
    assert grid.shape == (g["symbols"].shape[0], K * M) and np.array_equal(grid, pygfdm_grid(g))
    assert np.array_equal(mapper.map_to_resources(g["symbols"][1], ninput_size=nsym), grid[1])            # one block == row of the batch
    dem = gfdm_amd.ResourceMapper(M, K, A, g["smap"], True)                               # pygfdm demaps per timeslot only
    assert np.array_equal(dem.demap_from_resources(g["grid_in"]), c64(g["pygfdm_demapped"]))
    # the mapper's own inverse, both symbol orders, truncated output
    assert np.array_equal(mapper.demap_from_resources(grid, noutput_size=nsym), c64(g["symbols"]))
    assert np.array_equal(mapper.demap_from_resources(grid, noutput_size=7), c64(g["symbols"])[:, :7])
    for port, s in enumerate(g["shifts"]):
        pre = gfdm_amd.CyclicPrefixer(M * K, g["cp"], g["cs"], g["ramp"], g["window"], int(s))
        assert (pre.block_size(), pre.frame_size(), pre.cyclic_shift()) == (M * K, M * K + g["cp"] + g["cs"], int(s))
        frames = pre.add_cyclic_prefix(g["pygfdm_blocks"])
        assert rel_err(frames, g["pygfdm_frames"][port][:, plen:]) < 1e-6
        assert np.array_equal(pre.generic_work(g["pygfdm_blocks"][0]), frames[0])
        assert np.array_equal(pre.remove_cyclic_prefix(frames), frames[:, g["cp"]:g["cp"] + M * K])
        # only the 2 * ramp_len ramp taps given (lib/add_cyclic_prefix_cc.cc:42-56), shift passed per call
        short = gfdm_amd.CyclicPrefixer(M * K, g["cp"], g["cs"], g["ramp"], np.concatenate((g["window"][:g["ramp"]], g["window"][-g["ramp"]:])))
        assert np.array_equal(short.add_cyclic_prefix(g["pygfdm_blocks"], cyclic_shift=int(s)), frames)
    # the reference's Python classes (resource_mapper_python.cc, cyclic_prefix_python.cc): full blocks, one at a time
    if nsym == A * M:
        pm = gfdm_python.Resource_mapper(M, K, A, [int(x) for x in g["smap"]], g["per_timeslot"])
        assert (pm.block_size(), pm.frame_size()) == (A * M, K * M)
        assert np.array_equal(pm.map_to_resources(g["symbols"][0]), grid[0])
        assert np.array_equal(pm.map_to_resources(g["symbols"]), grid)                      # batch addition
        assert np.array_equal(pm.demap_from_resources(grid[0]), c64(g["symbols"][0]))
    pc = gfdm_python.Cyclic_prefixer(block_len=M * K, cp_len=g["cp"], cs_len=g["cs"], ramp_len=g["ramp"], window_taps=list(g["window"]),
                                     cyclic_shift=int(g["shifts"][-1]))
    assert (pc.block_size(), pc.frame_size(), pc.cyclic_shift()) == (M * K, M * K + g["cp"] + g["cs"], int(g["shifts"][-1]))
    assert np.array_equal(pc.add_cyclic_prefix(g["pygfdm_blocks"][0]), frames[0])
    assert np.array_equal(pc.remove_cyclic_prefix(frames[0]), frames[0][g["cp"]:g["cp"] + M * K])
    assert np.array_equal(pc.add_cyclic_prefix(g["pygfdm_blocks"]), frames)


@pytest.mark.parametrize("M,K,A,per_ts", [(9, 64, 52, True), (9, 64, 52, False), (5, 12, 7, True), (1, 8, 8, False), (31, 256, 200, True), (4, 3, 1, False)])
def test_mapper_against_oracle_ragged_sizes(M, K, A, per_ts):
    """every ninput_size / noutput_size from empty to full on random active sets, both symbol orders; device path == host path"""
    import torch
    import gfdm_amd
    rng = np.random.default_rng(M * K + A)
    smap = np.sort(rng.choice(K, A, replace=False))
    m = gfdm_amd.ResourceMapper(M, K, A, smap, per_ts)
    B = 5
    for n in sorted({1, 2, A - 1 or 1, A, A * M // 2 or 1, A * M - 1 or 1, A * M}):
        sym = c64(rng.standard_normal((B, n)) + 1j * rng.standard_normal((B, n)))
        grid = m.map_to_resources(sym, ninput_size=n)
        assert np.array_equal(grid, c64(R.map_to_resources(sym, M, K, smap, per_ts)))
        full = c64(rng.standard_normal((B, K * M)) + 1j * rng.standard_normal((B, K * M)))
        assert np.array_equal(m.demap_from_resources(full, noutput_size=n), c64(R.demap_from_resources(full, M, K, smap, per_ts, n)))
        dgrid = m.map_to_resources(torch.tensor(sym, device="cuda:0"), ninput_size=n)
        assert np.array_equal(dgrid.cpu().numpy(), grid)
        assert np.array_equal(m.demap_from_resources(torch.tensor(full, device="cuda:0"), noutput_size=n).cpu().numpy(),
                              m.demap_from_resources(full, noutput_size=n))


@pytest.mark.parametrize("N,cp,cs,ramp,shift", [(576, 16, 8, 8, 0), (576, 16, 8, 8, 8), (160, 0, 0, 0, 0), (64, 5, 3, 2, 1), (7936, 64, 32, 16, 9), (12, 12, 4, 14, 0)])
def test_prefixer_against_oracle(N, cp, cs, ramp, shift):
    import torch
    import gfdm_amd
    rng = np.random.default_rng(N + cp + shift)
    F = N + cp + cs
    window = rng.standard_normal(F) + 1j * rng.standard_normal(F)
    x = c64(rng.standard_normal((6, N)) + 1j * rng.standard_normal((6, N)))
    p = gfdm_amd.CyclicPrefixer(N, cp, cs, ramp, window, shift)
    ref = R.add_cyclic_prefix(x, cp, cs, ramp, c64(window), shift)
    got = p.add_cyclic_prefix(x)
    assert got.shape == (6, F) and rel_err(got, ref) < 1e-6
    if ramp < F:
        assert np.array_equal(got[:, ramp:F - ramp], c64(ref)[:, ramp:F - ramp])           # outside the ramps: a bit-exact copy
    assert np.array_equal(p.remove_cyclic_prefix(got), got[:, cp:cp + N])
    dev = p.add_cyclic_prefix(torch.tensor(x, device="cuda:0"))
    assert np.array_equal(dev.cpu().numpy(), got)
    assert np.array_equal(p.remove_cyclic_prefix(dev).cpu().numpy(), got[:, cp:cp + N])


def test_stage_round_trips_at_batch():
    """65 536 blocks of K=64 M=9 device resident: map -> demap and add prefix -> remove prefix give the input back; the mapped grid
    through the plain modulator equals the fused transmitter's modulate (same arithmetic, the grid just took the detour through HBM)."""
    import torch
    import gfdm_amd
    from gfdm_amd import synth
    from gfdm_amd.filters import get_frequency_domain_filter
    M, K, A, L, B = 9, 64, 52, 2, 65536
    dev = torch.device("cuda:0")
    smap = np.concatenate((np.arange(1, A // 2 + 1), np.arange(K - A // 2, K)))
    sym = synth.qpsk_symbols(0, B, A * M, dev)
    for per_ts in (True, False):
        m = gfdm_amd.ResourceMapper(M, K, A, smap, per_ts)
        grid = m.map_to_resources(sym)
        assert grid.shape == (B, K * M) and torch.equal(m.demap_from_resources(grid), sym)
        assert int((grid.abs() > 0).sum()) == B * A * M
    taps = get_frequency_domain_filter("rrc", 0.2, M, K, L)
    m = gfdm_amd.ResourceMapper(M, K, A, smap, True)
    tx = gfdm_amd.Transmitter(M, K, A, 16, 8, 0, smap, True, L, taps, np.zeros(0, complex), [0], [np.zeros(0, complex)])
    blocks = gfdm_amd.Modulator(M, K, L, taps).modulate(m.map_to_resources(sym[:4096]))
    assert torch.equal(blocks, tx.modulate(sym[:4096]))
    p = gfdm_amd.CyclicPrefixer(K * M, 16, 8, 0, np.zeros(0, complex), 0)
    frames = p.add_cyclic_prefix(blocks)
    assert torch.equal(frames, tx.add_frame(blocks, 0)) and torch.equal(p.remove_cyclic_prefix(frames), blocks)


def test_stage_argument_errors():
    """constructor / call errors with the reference's messages (lib/resource_mapper_kernel_cc.cc:44-69,78-82,95-99,
    lib/add_cyclic_prefix_cc.cc:42-50; python/bindings/resource_mapper_python.cc, cyclic_prefix_python.cc)"""
    import gfdm_amd
    import gfdm_python
    with pytest.raises(ValueError, match=r"active_subcarriers\(9\) MUST be smaller or equal to subcarriers\(8\)!"):
        gfdm_amd.ResourceMapper(4, 8, 9, np.arange(9))
    with pytest.raises(ValueError, match=r"number of subcarrier_map entries\(3\) MUST be equal to active_subcarriers\(4\)!"):
        gfdm_python.Resource_mapper(4, 8, 4, [0, 1, 2], True)
    with pytest.raises(ValueError, match="MUST be unique"):
        gfdm_amd.ResourceMapper(4, 8, 3, [1, 1, 2])
    with pytest.raises(ValueError, match="greater or equal to ZERO"):
        gfdm_amd.ResourceMapper(4, 8, 3, [-1, 1, 2])
    with pytest.raises(ValueError, match="MUST be smaller than subcarriers"):
        gfdm_amd.ResourceMapper(4, 8, 3, [1, 2, 8])
    m = gfdm_amd.ResourceMapper(4, 8, 3, [1, 2, 5])
    with pytest.raises(ValueError, match=r"input vector size\(13\) MUST not exceed active_subcarriers \* timeslots\(12\)!"):
        m.map_to_resources(np.zeros(13, np.complex64), ninput_size=13)
    with pytest.raises(ValueError, match=r"output vector size\(13\) MUST not exceed"):
        m.demap_from_resources(np.zeros(32, np.complex64), noutput_size=13)
    pm = gfdm_python.Resource_mapper(4, 8, 3, [1, 2, 5], True)
    with pytest.raises(RuntimeError, match=r"Input vector size\(11\) MUST be equal to Modulator.block_size\(12\)!"):
        pm.map_to_resources(np.zeros(11, np.complex64))
    with pytest.raises(RuntimeError, match=r"Input vector size\(31\) MUST be equal to Modulator.block_size\(32\)!"):
        pm.demap_from_resources(np.zeros(31, np.complex64))
    with pytest.raises(RuntimeError, match="Only ONE-dimensional vectors allowed!"):
        pm.map_to_resources(np.zeros((2, 2, 3), np.complex64))
    with pytest.raises(ValueError, match=r"number of window taps\(5\) MUST be equal to 2\*ramp_len\(4\) OR block_len\+cp_len \(14\)!"):
        gfdm_amd.CyclicPrefixer(8, 4, 2, 2, np.ones(5))
    with pytest.raises(ValueError, match="cyclic shift"):
        gfdm_amd.CyclicPrefixer(8, 4, 2, 2, np.ones(4), 3)
    # a suffix longer than the block: the reference's memcpy of in[0, cs - shift) reads past its input there (and so would a kernel)
    with pytest.raises(ValueError, match="cs_len - shift"):
        gfdm_amd.CyclicPrefixer(8, 0, 12, 0, np.zeros(0), 3)
    ok = gfdm_amd.CyclicPrefixer(8, 0, 12, 0, np.zeros(0), 4)               # cs - shift == block: the whole block repeats once
    assert np.array_equal(ok.add_cyclic_prefix(np.arange(8, dtype=np.complex64)),
                          R.add_cyclic_prefix(np.arange(8), 0, 12, 0, np.zeros(0), 4).astype(np.complex64))
    with pytest.raises(ValueError, match="cs_len - shift"):
        ok.add_cyclic_prefix(np.zeros(8, np.complex64), cyclic_shift=3)
    pc = gfdm_python.Cyclic_prefixer(8, 4, 2, 2, [1, 1, 1, 1])
    with pytest.raises(RuntimeError, match=r"Input vector size\(9\) MUST be equal to Cyclic_prefix.block_size\(8\)!"):
        pc.add_cyclic_prefix(np.zeros(9, np.complex64))
    with pytest.raises(RuntimeError, match=r"Input vector size\(9\) MUST be equal to Cyclic_prefix.frame_size\(8\)!"):
        pc.remove_cyclic_prefix(np.zeros(9, np.complex64))
    with pytest.raises(ValueError, match="cyclic shift"):
        gfdm_amd.CyclicPrefixer(8, 4, 2, 0, np.zeros(0)).add_cyclic_prefix(np.zeros(8, np.complex64), cyclic_shift=3)


@pytest.mark.parametrize("per_ts", [True, False])
def test_reference_mapper_test_shape_through_the_block_body(per_ts):
    """python/qa_resource_mapper_cc.py:38-88 / qa_resource_demapper_cc.py: 205 timeslots x 128 subcarriers, 110 active (centred map), three
    frames through the block -- here through the batched general_work body and the scheduler stand-in, expectation from the oracle
    (pinned to pygfdm's map_to_waveform_resources in tests/test_oracle.py)."""
    import gfdm_python
    import gfdm_testing as T
    M, K, A, frames = 205, 128, 110, 3
    smap = np.arange(A) + (K - A) // 2
    rng = np.random.default_rng(12)
    sym = (((1 - 2 * rng.integers(0, 2, (frames, A * M))) + 1j * (1 - 2 * rng.integers(0, 2, (frames, A * M)))) / np.sqrt(2)).astype(np.complex64)
    mapper = gfdm_python.Resource_mapper(M, K, A, smap.tolist(), per_ts)
    grid, produced = T.scheduler_run_mapper(mapper, True, sym.reshape(-1), [(2 * K * M + 5, 3 * A * M), (4 * K * M, A * M)])
    assert produced == [2, 1]
    assert np.array_equal(grid, R.map_to_resources(sym, M, K, smap, per_ts).astype(np.complex64))
    demapper = gfdm_python.Resource_mapper(M, K, A, smap.tolist(), per_ts, False)
    back, produced = T.scheduler_run_mapper(demapper, False, grid.reshape(-1), [(3 * A * M, 3 * K * M)])
    assert produced == [3] and np.array_equal(back, sym)


def test_reference_prefixer_test_cases():
    """python/qa_cyclic_prefixer_cc.py: :36-46 constructor accepts 2 * ramp_len or block_len + cp_len window taps and refuses
    block_len; :48-62 plain cyclic prefix under an all-ones window; :64-94 cyclic prefix + suffix pinched by a raised-cosine ramp
    (window built as pygfdm.cyclic_prefix.get_raised_cosine_ramp does: half-cosine flanks of ramp_len samples around ones)."""
    import gfdm_amd
    gfdm_amd.CyclicPrefixer(16 * 8, 4, 0, 4, np.arange(4 * 2))
    gfdm_amd.CyclicPrefixer(16 * 8, 4, 0, 4, np.arange(16 * 8 + 4))
    with pytest.raises(ValueError, match="number of window taps"):
        gfdm_amd.CyclicPrefixer(16 * 8, 4, 0, 4, np.arange(16 * 8))
    block_len, cp = 48, 8
    data = (np.arange(block_len) + 1).astype(np.complex64)
    got = gfdm_amd.CyclicPrefixer(block_len, cp, 0, 0, np.ones(block_len + cp)).add_cyclic_prefix(data)
    assert np.array_equal(got, np.concatenate((data[-cp:], data)))
    K, M, cp, ramp = 8, 8, 8, 4
    cs, N = 2 * ramp, 64
    F = N + cp + cs
    flank = 0.5 * (1.0 + np.cos(np.pi * (np.arange(ramp) + 0.5) / ramp + np.pi))          # rising half cosine, 0 .. 1 exclusive
    window = np.concatenate((flank, np.ones(F - 2 * ramp), flank[::-1]))
    data = (np.arange(N) + 1).astype(np.complex64)
    ref = np.concatenate((data[-cp:], data, data[:cs])) * window
    got = gfdm_amd.CyclicPrefixer(N, cp, cs, ramp, window).add_cyclic_prefix(np.tile(data, 3))
    assert got.shape == (3, F) and rel_err(got, np.tile(ref, (3, 1))) < 1e-6
