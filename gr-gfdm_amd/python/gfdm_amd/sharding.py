"""Batch sharding of independent GFDM blocks over the GPUs of one node.

GFDM blocks carry no state between them, so the batch is split contiguously: rank r of G processes blocks
[r*B/G, (r+1)*B/G).  There is NO payload collective on the data path; torch.distributed (RCCL over xGMI on the
GPU box, gloo in the CPU tests) is used only to agree on run statistics: blocks processed, an output checksum,
and the slowest rank's time.
"""
import torch
import torch.distributed as dist


def shard_range(total_blocks, rank, world_size):
    """Contiguous, balanced partition: the first (total % world) ranks get one extra block."""
    base, extra = divmod(total_blocks, world_size)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def output_checksum(out):
    """(sum re, sum im, sum |.|^2) in float64: a size-independent fingerprint of a result tensor."""
    o = out.reshape(-1)
    re = o.real.to(torch.float64)
    im = o.imag.to(torch.float64)
    return torch.stack([re.sum(), im.sum(), (re * re + im * im).sum()])


def reduce_stats(nblocks, checksum, elapsed_s, device):
    """All-reduce run statistics.  Returns (total blocks, summed checksum[3], max elapsed seconds).  With an initialised process group the
    two all-reduces run whatever the world size is (a one-rank RCCL group is how the collective path is exercised on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return int(nblocks), checksum.detach().to("cpu"), float(elapsed_s)
    if dist.get_backend() == "gloo":
        device = "cpu"                       # gloo reduces host tensors (the CPU tests; bench.py --dist-backend gloo)
    sums = torch.cat([torch.tensor([float(nblocks)], dtype=torch.float64, device=device), checksum.to(device)])
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    tmax = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return int(round(sums[0].item())), sums[1:].to("cpu"), float(tmax.item())


class ShardedBatch:
    """The batched-blocks multi-GPU mode: ONE batch of independent GFDM blocks over the GPUs of a node.

    `make_kernel(device)` builds the kernel object of one device (a gfdm_amd.Modulator / Demodulator / AdvancedReceiver / Transmitter
    created with device=...); this object then owns one kernel handle and one HIP stream per entry of `devices` and splits every batch
    contiguously with shard_range over world_size * len(devices) shards -- shard index rank * len(devices) + i belongs to local device i
    of process `rank`.  Two ways to use it:

      * one process per GPU (torchrun; bench.py): devices = [LOCAL_RANK], rank / world_size from the environment.  Every process
        launches its own shard; nothing is exchanged on the data path (torch.distributed carries reduce_stats only);
      * one process driving several GPUs: devices = [0, 1, ...], world_size = 1; `run` enqueues the shards on the devices' streams
        back to back (launches are asynchronous, so the devices work concurrently) and `synchronize` waits for all of them.

    There is no payload collective in either form: device d reads and writes only its own shard.
    """

    def __init__(self, make_kernel, devices, rank=0, world_size=1):
        self.devices = list(devices)
        if not self.devices:
            raise ValueError("ShardedBatch needs at least one device")
        if not (0 <= rank < world_size):
            raise ValueError("rank %d outside world of %d" % (rank, world_size))
        self.rank, self.world_size = int(rank), int(world_size)
        self.kernels = [make_kernel(d) for d in self.devices]
        self.streams = [self._make_stream(d) for d in self.devices]

    @staticmethod
    def _make_stream(device):
        if isinstance(device, str) and device == "cpu":          # CPU stand-in kernels of the gloo tests: no stream
            return None
        return torch.cuda.Stream(device=device)

    @property
    def n_shards(self):
        return self.world_size * len(self.devices)

    def shard(self, total_blocks, i=0):
        """(first block, number of blocks) of local device i's shard of a batch of total_blocks"""
        return shard_range(total_blocks, self.rank * len(self.devices) + i, self.n_shards)

    def plan(self, total_blocks):
        """[(device, first block, number of blocks)] for this process"""
        return [(d,) + self.shard(total_blocks, i) for i, d in enumerate(self.devices)]

    def local_blocks(self, total_blocks):
        return sum(n for _, _, n in self.plan(total_blocks))

    def stream_ptr(self, i=0):
        return 0 if self.streams[i] is None else self.streams[i].cuda_stream

    def prepare(self, c_function, outs, ins, counts, stream_ptrs=None):
        """Pre-marshalled launch of the C-ABI entry point `c_function` (e.g. lib().gfdm_hip_advanced_receiver_work_device; signature
        (handle, out, in..., nblocks, stream)) for every local device: outs[i] / ins[i] are that device's resident tensors (ins[i] a
        tuple, None entries allowed), counts[i] its number of blocks.  Returns a callable that enqueues all of them (about a microsecond
        of host time per launch) on the devices' own streams, or on stream_ptrs[i] when given."""
        import ctypes
        calls = []
        for i, k in enumerate(self.kernels):
            sp = self.stream_ptr(i) if stream_ptrs is None else stream_ptrs[i]
            args = [k._h, ctypes.c_void_p(outs[i].data_ptr())]
            args += [ctypes.c_void_p(t.data_ptr()) if t is not None else None for t in ins[i]]
            args += [ctypes.c_int64(int(counts[i])), ctypes.c_void_p(sp)]
            calls.append(args)

        def go():
            for args in calls:
                rc = c_function(*args)
                if rc != 0:
                    raise RuntimeError("gfdm_hip launch failed: %d" % rc)
        return go

    def run(self, method, total_blocks, ins, outs=None):
        """kernel.<method>(*ins[i], out=outs[i], stream=...) for every local device, ins[i] being the device-resident inputs of ITS shard
        (shard(total_blocks, i) blocks each).  Asynchronous; returns the per-device results."""
        res = []
        for i, k in enumerate(self.kernels):
            _, n = self.shard(total_blocks, i)
            kw = {}
            if outs is not None:
                kw["out"] = outs[i]
            if self.streams[i] is not None:
                kw["stream"] = self.streams[i]
            res.append(getattr(k, method)(*ins[i], **kw) if n else None)
        return res

    def run_global(self, method, global_ins, block_elems):
        """Host batches in, host result out: every global input (a numpy array of total_blocks * block_elems[j] elements) is sliced to the
        local shards, handed to the kernels' host entry points (gfdm_hostpipe: bounce in chunks or in place) and the local results are
        returned with their block range: [(first block, number of blocks, result)].  The host entry points block until their result is
        back, so every local device gets a host thread of its own (ctypes releases the GIL for the duration of the C call) -- the Python
        twin of gfdm/sharded_batch.h; the first exception of any shard is re-raised after all threads have finished.  The caller owns
        assembling ranks' results, if it wants them at all.  gfdm_amd.host_call_stats() is per calling thread, so each shard's statistics
        are read on ITS thread and kept in self.last_host_call_stats (one dict per local shard that had blocks)."""
        import threading
        import numpy as np
        first = np.asarray(global_ins[0])
        total = first.size // block_elems[0]
        work = []
        for i, k in enumerate(self.kernels):
            s, n = self.shard(total, i)
            if n:
                work.append((k, s, n, [np.asarray(g).reshape(total, -1)[s:s + n] for g in global_ins]))
        results, errors, stats = [None] * len(work), [None] * len(work), [None] * len(work)

        def shard_call(j):
            k, s, n, parts = work[j]
            try:
                results[j] = (s, n, getattr(k, method)(*parts))
                if hasattr(k, "_h"):         # a gfdm_amd kernel object (the gloo tests use CPU stand-ins)
                    from . import capi
                    stats[j] = capi.host_call_stats()
            except BaseException as e:       # re-raised on the calling thread
                errors[j] = e

        if len(work) == 1:
            shard_call(0)
        else:
            threads = [threading.Thread(target=shard_call, args=(j,), name="gfdm-shard-%d" % j) for j in range(len(work))]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        for e in errors:
            if e is not None:
                raise e
        self.last_host_call_stats = stats
        return results

    def synchronize(self):
        for st in self.streams:
            if st is not None:
                st.synchronize()
