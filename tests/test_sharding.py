"""Multi-GPU batch sharding, exercised on CPU with world_size 2 over gloo.  The compute step is played by the
oracle here (tests may use it); on the GPU box the same helpers wrap the HIP kernels (bench.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gfdm_ref as R
from gfdm_amd import sharding, synth
from gfdm_amd.filters import get_frequency_domain_filter


def test_shard_range_is_a_partition():
    for total in (0, 1, 7, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == total
            for (s0, n0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + n0 == s1
            assert max(n for _, n in spans) - min(n for _, n in spans) <= 1


def test_synthetic_inputs_do_not_depend_on_the_split():
    whole = synth.qpsk_symbols(0, 10, 160, "cpu")
    parts = [synth.qpsk_symbols(*sharding.shard_range(10, r, 3), 160, "cpu") for r in range(3)]
    assert torch.equal(whole, torch.cat(parts))
    assert torch.allclose(whole.abs(), torch.ones(()))
    f = synth.channel_response(3, 2, 160, "cpu")
    assert torch.allclose(f, synth.channel_response(0, 5, 160, "cpu")[3:])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    M, K, L = 5, 32, 2
    taps = R.normalize_taps(get_frequency_domain_filter("rrc", 0.5, M, K, L), M)
    start, n = sharding.shard_range(total, rank, world)
    sym = synth.qpsk_symbols(start, n, M * K, "cpu").numpy()
    out = R.demodulate(R.modulate(sym, taps, M, K, L), taps, M, K, L).astype(np.complex64)
    blocks, chk, tmax = sharding.reduce_stats(n, sharding.output_checksum(torch.from_numpy(out)), 0.1 * (rank + 1), "cpu")
    q.put((rank, blocks, chk.numpy(), tmax))
    dist.destroy_process_group()


def test_two_rank_shard_matches_single_process():
    total, world = 9, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    M, K, L = 5, 32, 2
    taps = R.normalize_taps(get_frequency_domain_filter("rrc", 0.5, M, K, L), M)
    sym = synth.qpsk_symbols(0, total, M * K, "cpu").numpy()
    out = R.demodulate(R.modulate(sym, taps, M, K, L), taps, M, K, L).astype(np.complex64)
    ref = sharding.output_checksum(torch.from_numpy(out)).numpy()
    for rank, blocks, chk, tmax in results:
        assert blocks == total
        assert np.allclose(chk, ref, rtol=1e-9, atol=1e-6)
        assert abs(tmax - 0.2) < 1e-12              # max over ranks


# ---- the product entry point of the batched-blocks multi-GPU mode: gfdm_amd.sharding.ShardedBatch ----------------------------------------
# (the compute of a device is played by the oracle here: a stand-in with the method names of gfdm_amd.AdvancedReceiver; on the GPU box
# tests/test_parity_gpu.py::test_sharded_batch_on_the_gpu runs the real kernel objects through the same class)

class _OracleReceiver:
    M, K, L = 5, 32, 2

    def __init__(self, device):
        self.device = device
        self.taps = R.normalize_taps(get_frequency_domain_filter("rrc", 0.5, self.M, self.K, self.L), self.M)

    def demodulate_equalize(self, x, f_eq):
        return R.advanced_receive(x, self.taps, self.M, self.K, self.L, np.arange(self.K), R.qpsk_points(), 2, f_eq=f_eq, kind="qpsk").astype(np.complex64)


def _sharded_inputs(total):
    M, K, L = _OracleReceiver.M, _OracleReceiver.K, _OracleReceiver.L
    taps = R.normalize_taps(get_frequency_domain_filter("rrc", 0.5, M, K, L), M)
    sym = synth.qpsk_symbols(0, total, M * K, "cpu").numpy()
    feq = synth.channel_response(0, total, M * K, "cpu").numpy()
    x = np.fft.ifft(np.fft.fft(R.modulate(sym, taps, M, K, L), axis=-1) * feq, axis=-1)
    return x, feq


def test_sharded_batch_plan_is_a_partition_over_ranks_and_devices():
    for total in (0, 5, 4096, 65537):
        for world, ndev in ((1, 1), (1, 8), (2, 4), (8, 1), (3, 2)):
            spans = []
            for r in range(world):
                sb = sharding.ShardedBatch(lambda d: object(), ["cpu"] * ndev, rank=r, world_size=world)
                assert sb.n_shards == world * ndev
                spans += [(s, n) for _, s, n in sb.plan(total)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == total
            for (s0, n0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + n0 == s1
    try:
        sharding.ShardedBatch(lambda d: object(), [])
        assert False
    except ValueError:
        pass


def _sharded_worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, feq = _sharded_inputs(total)                        # every rank holds the description of the whole batch, computes only its shards
    sb = sharding.ShardedBatch(_OracleReceiver, ["cpu", "cpu"], rank=rank, world_size=world)
    parts = sb.run_global("demodulate_equalize", [x, feq], [x.shape[1], x.shape[1]])
    sb.synchronize()
    local = np.concatenate([p for _, _, p in parts]) if parts else np.zeros((0, x.shape[1]), np.complex64)
    blocks, chk, _ = sharding.reduce_stats(sb.local_blocks(total), sharding.output_checksum(torch.from_numpy(local)), 0.0, "cpu")
    q.put((rank, [(s, n) for s, n, _ in parts], local, blocks, chk.numpy()))
    dist.destroy_process_group()


def test_sharded_batch_two_ranks_two_devices_each_match_single_process():
    total, world = 11, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, feq = _sharded_inputs(total)
    whole = _OracleReceiver("cpu").demodulate_equalize(x, feq)
    ref_chk = sharding.output_checksum(torch.from_numpy(whole)).numpy()
    assert [sp for _, spans, _, _, _ in results for sp in spans] == [(0, 3), (3, 3), (6, 3), (9, 2)]      # 4 shards, contiguous, balanced
    assert np.array_equal(np.concatenate([loc for _, _, loc, _, _ in results]), whole)                   # no block lost, none computed twice
    for _, _, _, blocks, chk in results:
        assert blocks == total and np.allclose(chk, ref_chk, rtol=1e-9, atol=1e-6)
