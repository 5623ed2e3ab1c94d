"""profiles/README.md for round 5: every row of profiles/r05/kernel_alone.csv with its roofline fraction, the HBM traffic ratios, the file index."""
import csv, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r05")
BPS = {"modulate": 16, "demod_mf": 16, "demod_zf": 24, "demod_mf_ic2": 16, "demod_zf_ic2": 24}
out = ["# profiles/ -- measured evidence", "",
       "Round 5 (library build `%s`, one MI355X per collection; the boxes of the pool differ by ~3 %%).  Older rounds: `r01/` ... `r04/` (`r04/README.md` = the round-4 tables)," % open(os.path.join(P, "build_id.txt")).read().strip(),
       "`EXPERIMENTS.md` = what was tried and dropped, `DESIGN_r04_long.md` = the long form of DESIGN.md up to round 4.", "",
       "## rocprofv3 --kernel-trace, one kernel on the GPU at a time (`r05/kernel_alone.csv`, `scratch/gpu_r5.sh alone`)", "",
       "Algorithmic bytes per block: 16 N (modulate, MF demod, with or without IC), 24 N (with the per-block equaliser vector); peak 8 TB/s.", "",
       "| K, M, L | path | blocks / launch | kernel | launches | mean us | median us | GB/s | % of peak |", "|---|---|---|---|---|---|---|---|---|"]
for r in csv.DictReader(open(os.path.join(P, "kernel_alone.csv"))):
    parts = r["label"].split("_")
    K, M, L, B = int(parts[0]), int(parts[1]), int(parts[2]), int(parts[-1])
    path = "_".join(parts[3:-1])
    dense = path.startswith("dense_")
    base = path.replace("dense_", "")
    nbytes = BPS[base] * K * M * B
    gbps = nbytes / (float(r["mean_us"]) * 1e-6) / 1e9
    out.append("| %d, %d, %d | %s%s | %d | `%s` | %s | %.2f | %.2f | %.0f | **%.1f** |" % (K, M, L, base, " (dense matrix-core transforms, `set_dft_matrix_cores(2)`)" if dense else "", B, r["kernel"], r["launches"],
                                                                                  float(r["mean_us"]), float(r["median_us"]), gbps, 100 * gbps / 8000))
out += ["", "## HBM traffic (`r05/pmc_hbm_traffic_summary.csv`: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md)", "",
        "| run | kernel | traffic per launch | algorithmic bytes | ratio |", "|---|---|---|---|---|"]
acc = {}
for r in csv.DictReader(open(os.path.join(P, "pmc_hbm_traffic_summary.csv"))):
    parts = r["run"].split("_")
    B, K, M = int(parts[-1]), int(parts[-4]), int(parts[-3])
    path = "_".join(parts[:-4])
    if ("modulate" in r["kernel"]) != (path == "modulate"):
        continue
    acc.setdefault((r["run"], r["kernel"], BPS[path] * K * M * B), {})[r["counter"]] = float(r["mean_KiB"])
for (run, kern, alg), v in sorted(acc.items()):
    t = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    out.append("| %s | `%s` | %.1f MB | %.1f MB | %.4f |" % (run, kern, t / 1e6, alg / 1e6, t / alg))
out += ["", "## Files of round 5", "",
        "| file | what |", "|---|---|",
        "| `r05/bench_default.json`, `bench_driver_line.json`, `bench_cfg3.json`, `bench_cfg4.json`, `bench_cfg5.json` | `bench.py` lines of the final build (default; the driver's `--gpus 1 --steps 20 --warmup 5`; `--config cfg3/4/5`): `roofline` carries the per-launch, burst, sustained and rocprofv3 readings and `north_star` |",
        "| `r05/before_the_roofline_fix/bench_driver_line.json` | the same line before the per-launch pairs got their warm-up run: 20 pairs behind an idle gap read 15.1 us (0.31) for a kernel every other reading puts at 9.2-11.6 us |",
        "| `r05/sustained_vs_burst.txt`, `r05/sustained_probe/` | why one kernel read 0.54 back to back and 0.69 isolated in round 4: steady state vs bursts, shader clock launch by launch, copy control |",
        "| `r05/kernel_alone.csv`, `r05/pmc_hbm_traffic_summary.csv`, `r05/build_id.txt` | the tables above; `bench.py` reads them for `kernel_ms_rocprofv3` / `traffic` when the build id matches |",
        "| `r05/pytest_gpu_suite.txt` | the `-m gpu` suite on the final tree |",
        "| `r05/fuzz_final_build.txt` | `scratch/fuzz_host_path.py` (72 930 random host calls over 12 shapes incl. the Rader shape with its cancellation rounds, every result equal to the device path) and `scratch/fuzz_shapes.py` (45 random shapes on the tuned / run-time instantiated families, 8 651 on the generic family; no failure, worst weighted error 4.3e-6 against the bound of 1e-5 -- the prefixer term counts tenfold there) on the final build |", ""]
open(os.path.join(ROOT, "profiles", "README.md"), "w").write("\n".join(out))
print("\n".join(out[:14]))
