"""profiles/README.md of the current round (6): every row of profiles/r06/kernel_alone.csv with its roofline fraction, the HBM traffic ratios, the vector-ALU occupancy
from the SQ counters, the file index.  (Round 5's table: git show 27c6ffa:profiles/README.md.)"""
import csv, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = "r06"
P = os.path.join(ROOT, "profiles", RND)
BPS = {"modulate": 16, "demod_mf": 16, "demod_zf": 24, "demod_mf_ic2": 16, "demod_zf_ic2": 24}
out = ["# profiles/ -- measured evidence", "",
       "Round 6 (library build `%s`, one MI355X per collection; the boxes of the pool differ by ~3 %%).  Older rounds: `r01/` ... `r05/` (`r04/README.md` = the round-4 tables)," % open(os.path.join(P, "build_id.txt")).read().strip(),
       "`EXPERIMENTS.md` = what was tried and dropped, `DESIGN_r04_long.md` = the long form of DESIGN.md up to round 4.", "",
       "## rocprofv3 --kernel-trace, one kernel on the GPU at a time (`r06/kernel_alone.csv`, `scratch/gpu_r6.sh alone`)", "",
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
out += ["", "## HBM traffic (`r06/pmc_hbm_traffic_summary.csv`: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md)", "",
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
# ---- vector-ALU issue time from the SQ counters (quad-cycle units: SQ_ACTIVE_INST_VALU x 4 clocks, summed over the chip's 1024 SIMDs) against the launch duration above
dur = {}
for r in csv.DictReader(open(os.path.join(P, "kernel_alone.csv"))):
    parts = r["label"].split("_")
    if "dense" in parts:
        continue
    dur[("_".join(parts[3:-1]), int(parts[0]), int(parts[1]), int(parts[2]), int(parts[-1]), r["kernel"])] = float(r["mean_us"])
sq = {}
for r in csv.DictReader(open(os.path.join(P, "pmc_sq_counters_summary.csv"))):
    sq.setdefault((r["run"], r["kernel"]), {})[r["counter"]] = float(r["mean"])
out += ["", "## Vector work (`r06/pmc_sq_counters_summary.csv`, `scratch/gpu_r6.sh sq`; SQ_* cycle counters tick once per four clocks)", "",
        "Wave issue = SQ_ACTIVE_INST_VALU x 4 clocks / 1024 SIMDs / (rocprofv3 duration x 2.4 GHz): the clocks the waves of a SIMD spend issuing vector instructions (one per 4 clocks and wave), over the launch;",
        "two waves can issue side by side, so this is NOT the pipe's utilisation.  Pipe (lower bound) = vector instructions x 2 clocks (what the SIMD needs for a plain wave64 f32 instruction,",
        "`MI355X_MICROARCH.md`; packed f32, DPP and lane swaps cost 4.3-8.2, `scratch/probe/pk_rate.hip`) over the same.  Resident waves = SQ_WAVE_CYCLES x 4 / (GRBM_GUI_ACTIVE / 8) / 1024.", "",
        "| run | kernel | vector instructions / wave | launch us | wave issue / launch | pipe, lower bound | resident waves / SIMD | LDS bank-conflict cycles / busy LDS cycles |", "|---|---|---|---|---|---|---|---|"]
for (run, kern), v in sorted(sq.items()):
    parts = run.split("_")
    B, K, M, L = int(parts[-1]), int(parts[-4]), int(parts[-3]), int(parts[-2])
    path = "_".join(parts[:-4])
    us = dur.get((path, K, M, L, B, kern))
    if us is None or "SQ_ACTIVE_INST_VALU" not in v:
        continue
    clk = v["SQ_ACTIVE_INST_VALU"] * 4 / 1024
    pipe = v["SQ_INSTS_VALU"] * 2 / 1024
    res = v["SQ_WAVE_CYCLES"] * 4 / (v["GRBM_GUI_ACTIVE"] / 8) / 1024 if v.get("GRBM_GUI_ACTIVE") else float("nan")
    out.append("| %s | `%s` | %.0f | %.2f | %.0f %% | **%.0f %%** | %.1f | %.2f |" % (run, kern, v["SQ_INSTS_VALU"] / v["SQ_WAVES"], us, 100 * clk / (us * 2400), 100 * pipe / (us * 2400), res,
                                                                           v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_ACTIVE_INST_LDS"] * 4, 1)))
out += ["", "## Files of round 6", "",
        "| file | what |", "|---|---|",
        "| `r06/sanitizers_host_side.txt` | `make -C tests/sanitize run`: the host-side code (C-ABI, host-buffer path, run-time instantiation, C++ classes, sharded batch) under ThreadSanitizer and AddressSanitizer + UBSan against the loop-back HIP layer, 20 s each, clean; what the first runs found is in `tests/test_sanitizers.py` |",
        "| `r06/ic_valu_waves_ab.csv` | the ONE experiment on the K=64 M=9 cancellation kernels: a register bound for seven / eight waves per SIMD against the unbounded build (six), three alternating collections on one box -- slower (DESIGN.md section 7) |",
        "| `r06/valu_issue_costs.txt` | `scratch/probe/pk_rate.hip` on the box of the final collection: clocks per wave-instruction of plain / packed f32, DPP moves and lane swaps at 4 and 16 waves per CU (2.4-2.7 / 4.8 / 4.3 / 8.3) -- what the pipe estimates of DESIGN.md section 7 are priced with |",
        "| `r06/fuzz_host_path_soak.txt` | 15 minutes of `scratch/fuzz_host_path.py` on the GPU (187 600 random host calls, every result equal to the device path) |",
        "| `r06/ic_dpp_fold_ab.csv`, `r06/ic_hot_path_only_ab.csv` | same-box A/Bs behind DESIGN.md section 7: the DPP rotates folded into their adds (kept, -2.6 %), and the IC kernels without their cold paths = eight / seven waves per SIMD without spills (timing only: <= 1.5 % at 65 536 blocks, not kept) |",
        "| `r06/pmc_sq_counters_summary.csv` | SQ counters of the K=64 M=9 MF / MF + 2 IC / ZF + 2 IC kernels at 65 536 and 4096 blocks and of the Rader kernels (the table above) |",
        "| `r06/kernel_alone.csv`, `r06/pmc_hbm_traffic_summary.csv`, `r06/build_id.txt` | the tables above; `bench.py` reads them for `kernel_ms_rocprofv3` / `traffic` when the build id matches |",
        "| `r06/bench_default.json`, `bench_driver_line.json`, `bench_cfg3.json`, `bench_cfg4.json`, `bench_cfg5.json` | `bench.py` lines of the final build; `roofline` now carries the north-star kernel's readings as scalar keys (`north_star_frac*`) and `value_sustained` |",
        "| `r06/sticky_error_before_after.txt` | the sanitizer harness's first finding on the real runtime: after an application's failed `hipMalloc` the pre-fix library fails its next launch with `kernel launch: out of memory`, the fixed one does not |",
        "| `r06/pytest_gpu_suite.txt`, `r06/fuzz_final_build.txt` | the `-m gpu` suite and the two GPU fuzzers on the final tree |", ""]
open(os.path.join(ROOT, "profiles", "README.md"), "w").write("\n".join(out))
print("\n".join(out[:14]))
