"""Stand-alone mapper / demapper / cyclic prefixer kernels: event-timed duration and algorithmic bandwidth.
   python3 scratch/bench_stages.py [K M A B ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import numpy as np, torch
import gfdm_amd
dev = torch.device("cuda:0")
args = [int(a) for a in sys.argv[1:]] or [64, 9, 52, 4096, 64, 9, 52, 65536, 256, 31, 200, 8192]
for i in range(0, len(args), 4):
    K, M, A, B = args[i:i + 4]
    N, cp, cs = K * M, K // 4, K // 8
    smap = np.concatenate((np.arange(1, A // 2 + 1), np.arange(K - A // 2, K)))
    slots = max(2, min(8, int(2e9 // (B * N * 8 * 2))))
    syms = [torch.randn(B, A * M, dtype=torch.complex64, device=dev) for _ in range(slots)]
    grids = [torch.randn(B, N, dtype=torch.complex64, device=dev) for _ in range(slots)]
    F = N + cp + cs
    frames = [torch.randn(B, F, dtype=torch.complex64, device=dev) for _ in range(slots)]
    o_sym, o_grid, o_frame = torch.empty_like(syms[0]), torch.empty_like(grids[0]), torch.empty_like(frames[0])
    for per_ts in (True, False):
        m = gfdm_amd.ResourceMapper(M, K, A, smap, per_ts)
        p = gfdm_amd.CyclicPrefixer(N, cp, cs, cs, np.ones(F), 0)
        paths = {"map (per %s)" % ("timeslot" if per_ts else "subcarrier"): (8 * (A * M + N), lambda s: m.map_to_resources(syms[s], out=o_grid)),
                 "demap (per %s)" % ("timeslot" if per_ts else "subcarrier"): (16 * A * M, lambda s: m.demap_from_resources(grids[s], out=o_sym))}
        if per_ts:
            paths["add cyclic prefix"] = (8 * (N + F), lambda s: p.add_cyclic_prefix(grids[s], out=o_frame))
            paths["remove cyclic prefix"] = (16 * N, lambda s: p.remove_cyclic_prefix(frames[s], out=o_grid))
        for name, (bpb, fn) in paths.items():
            for r in range(3): fn(r % slots)
            torch.cuda.synchronize()
            reps = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(reps): fn(r % slots)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            print("K=%d M=%d A=%d B=%d  %-26s %8.1f us  %6.0f GB/s  %4.1f %% of 8 TB/s (launch to launch)" % (
                K, M, A, B, name, us, bpb * B / us / 1e3, bpb * B / us / 1e3 / 80))
