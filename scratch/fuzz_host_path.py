#!/usr/bin/env python3
"""Fuzz of the host-buffer batch path: random shapes (compiled, run-time instantiated, generic), entry points, block counts, routes, chunk sizes, staging depths, copy
threads, stream counts, registered / pageable / partly registered operands -- every result must EQUAL the device-pointer path's.  Not part of the suite (minutes):
    python3 scratch/fuzz_host_path.py [seconds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import gfdm_amd as g
import gfdm_ref as R
from gfdm_amd.filters import get_frequency_domain_filter
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "4")))
SHAPES = [(9, 64, 2), (5, 32, 2), (15, 128, 4), (31, 256, 2), (9, 128, 2), (7, 12, 2), (6, 16, 2), (21, 37, 2), (25, 96, 2), (16, 4, 2), (5, 64, 2), (127, 16, 2)]   # (127, 16: the Rader kernels)
g.set_jit(g.JIT_IN_CONSTRUCTOR)
handles = {}
def get(shape):
    if shape not in handles:
        M, K, L = shape
        taps = get_frequency_domain_filter("rrc", 0.3, M, K, L)
        handles[shape] = (g.Modulator(M, K, L, taps), g.Demodulator(M, K, L, taps), g.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, R.qpsk_points()))
    return handles[shape]
def dev(call, *arrs):
    t = [None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs]
    o = call(*t); torch.cuda.synchronize(); return o.cpu().numpy()
t0, n, routes = time.time(), 0, {}
while time.time() - t0 < budget:
    shape = SHAPES[rng.integers(len(SHAPES))]
    M, K, L = shape; N = M * K
    mod, dem, adv = get(shape)
    nb = int(rng.choice([1, 2, 3, 5, 8, 13, 33, 64, 100, 257, 600])) if N < 4000 else int(rng.choice([1, 2, 5, 17, 40]))
    # page-aligned arrays that own their pages: what gfdm_hip_register_host takes.  (The first version of this fuzz registered ordinary numpy arrays: small
    # ones live in heap pages they share with other objects, and pinning / unpinning such pages ended in GPU faults inside LATER, unrelated pageable copies
    # -- profiles/r04/host_path_fuzz.txt -- which is why the library now insists on whole pages.)
    mk = g.aligned_copy
    sym = mk((((1 - 2 * rng.integers(0, 2, (nb, N))) + 1j * (1 - 2 * rng.integers(0, 2, (nb, N)))) / np.sqrt(2)).astype(np.complex64))
    feq = mk((1.0 + 0.3 * (rng.standard_normal((nb, N)) + 1j * rng.standard_normal((nb, N)))).astype(np.complex64))
    frames = mk(dev(lambda s: mod.modulate(s), sym))
    if os.environ.get("FUZZ_VERBOSE"): print("   inputs ready", flush=True)
    which = int(rng.integers(5))
    ins, call, ref = [(sym,), (frames,), (frames, feq), (frames,), (frames, feq)][which], None, None
    fns = [lambda o: mod.modulate(sym, out=o), lambda o: dem.demodulate(frames, out=o), lambda o: dem.demodulate_equalize(frames, feq, out=o),
           lambda o: adv.demodulate(frames, out=o), lambda o: adv.demodulate_equalize(frames, feq, out=o)]
    refs = [lambda: frames, lambda: dev(lambda x: dem.demodulate(x), frames), lambda: dev(lambda x, e: dem.demodulate_equalize(x, e), frames, feq),
            lambda: dev(lambda x: adv.demodulate(x), frames), lambda: dev(lambda x, e: adv.demodulate_equalize(x, e), frames, feq)]
    ref = refs[which]()
    if os.environ.get("FUZZ_VERBOSE"): print("   reference ready", flush=True)
    mode = int(rng.integers(4)); chunk = int(rng.choice([0, 1, 8 * N, 3 * 24 * N + 5, 1 << 20, 1 << 30])); depth = int(rng.integers(1, 5)); thr = int(rng.integers(0, 4)); streams = int(rng.integers(1, 3))
    g.set_host_pipeline(mode, chunk, depth, thr, streams)
    out = mk(np.full((nb, N), np.nan + 0j, np.complex64))
    reg = [a for a, pick in zip((out,) + ins, rng.integers(0, 2, 1 + len(ins))) if pick]
    if os.environ.get("FUZZ_VERBOSE"):
        print(n, shape, nb, which, mode, chunk, depth, thr, streams, [hex(a.ctypes.data) + "+" + str(a.nbytes) for a in reg], [hex(a.ctypes.data) for a in (out,) + ins], flush=True)
    for a in reg: g.register_host(a)
    V = os.environ.get("FUZZ_VERBOSE")
    try:
        fns[which](out)
        if V: print("   call done", g.host_call_stats(), flush=True)
    finally:
        for a in reg: g.unregister_host(a)
    if V: print("   unregistered", flush=True)
    st = g.host_call_stats()
    assert np.array_equal(out, ref), (shape, nb, which, mode, chunk, depth, thr, streams, len(reg), st)
    routes[(mode, bin(st["direct_mask"]).count("1"), st["chunks"] > 1)] = routes.get((mode, bin(st["direct_mask"]).count("1"), st["chunks"] > 1), 0) + 1
    n += 1
print("host-path fuzz: %d random calls over %d shapes, every result equal to the device path; (route, operands in place, chunked) -> calls:" % (n, len(handles)))
for k in sorted(routes): print("  ", k, routes[k])
