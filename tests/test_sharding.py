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
