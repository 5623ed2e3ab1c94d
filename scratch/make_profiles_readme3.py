"""profiles/r03/* -> the measured tables of the r03 section of profiles/README.md (stdout)."""
import csv, json, os, sys
D = os.path.join("profiles", sys.argv[1] if len(sys.argv) > 1 else "r03")
rows = lambda n: list(csv.DictReader(open(os.path.join(D, n))))
out = []
w = out.append

LABEL = {"modulate": "modulate", "demod_mf": "MF demod", "demod_zf": "ZF demod", "demod_mf_ic2": "MF + 2 IC", "demod_zf_ic2": "ZF + 2 IC",
         "demod_mf_ic2_mx": "MF + 2 IC, rounds forced onto the matrix cores", "demod_zf_ic2_mx": "ZF + 2 IC, rounds forced onto the matrix cores",
         "demod_mf_ic2_valu": "MF + 2 IC, rounds on the vector ALU (`set_ic_matrix_cores(0)`)", "demod_zf_ic2_valu": "ZF + 2 IC, rounds on the vector ALU",
         "frames_zf_ic2_est": "estimator + ZF + 2 IC + demapper (52 active), one kernel", "estimate_frame": "estimate_frame"}
# round-2 figures of the same table (profiles/r02, another box of the pool): mean us
R2 = {("modulate", 4096): 10.17, ("demod_mf", 4096): 9.84, ("demod_zf", 4096): 12.80, ("demod_mf_ic2", 4096): 12.06, ("demod_zf_ic2", 4096): 14.70,
      ("frames_zf_ic2_est", 4096): 15.53, ("estimate_frame", 4096): 7.56, ("modulate", 65536): 107.57, ("demod_mf", 65536): 108.02, ("demod_zf", 65536): 147.03,
      ("demod_mf_ic2", 65536): 141.39, ("demod_zf_ic2", 65536): 167.93, ("frames_zf_ic2_est", 65536): 182.73, ("estimate_frame", 65536): 69.74}
R2S = {}
try:
    for _r in csv.DictReader(open(os.path.join("profiles", "r02", "shape_kernel_durations.csv"))):
        R2S[(_r["shape_batch"], _r["path"])] = float(_r["mean_us"])
except OSError:
    pass


def base_path(path):
    return path.replace("_mx", "").replace("_valu", "")


def bytes_per_block(path, K, M, A=52):
    N = K * M
    return {"modulate": 16 * N, "demod_mf": 16 * N, "demod_mf_ic2": 16 * N, "demod_zf": 24 * N, "demod_zf_ic2": 24 * N,
            "frames_zf_ic2_est": 8 * N + 16 * K + 8 * A * M, "estimate_frame": 16 * K + 8 * N}[base_path(path)]


w("### Kernel durations, K=64 M=9 L=2, ONE kernel on the GPU at a time (`kernel_alone_64_9_2.csv`)\n")
w("`rocprofv3 --kernel-trace -- python3 scratch/run_kernel.py <path> <blocks> <launches> <ring slots>` (begin/end timestamps; template arguments: K, M, L,")
w("mode (0 frequency-domain output, 1 demodulate, 2 IC), equaliser (0 none, 1 vector, 2 estimated in the kernel), IC rounds (0 general / none, 1 real even kernel on")
w("the vector ALU, 2 matrix cores)).  The boxes of the pool differ by ~10 % on the latency-bound 4096-block launches; the round-2 column is another box.\n")
w("| path | kernel | blocks / launch | launches | rocprofv3 mean (median, min) | algorithmic bytes / launch | achieved (mean) | of 8 TB/s | round 2, other box (mean) |")
w("|---|---|---|---|---|---|---|---|---|")
for r in rows("kernel_alone_64_9_2.csv"):
    B, path = int(r["batch"]), r["path"]
    byt = bytes_per_block(path, 64, 9) * B
    gb = byt / (float(r["mean_us"]) * 1e-6) / 1e9
    r2 = R2.get((path, B))
    w("| %s | `%s` | %d | %s | %.2f us (%.2f, %.2f) | %s | %.2f TB/s | **%.1f %%** | %s |" % (LABEL[path], r["kernel"], B, r["launches"], float(r["mean_us"]), float(r["median_us"]),
      float(r["min_us"]), format(byt, ","), gb / 1e3, gb / 80, ("%.2f us = %.1f %%" % (r2, byt / (r2 * 1e-6) / 8e10)) if r2 else ""))
w("")
w("### BASELINE configs[3] / [4] shapes at their per-GPU batch and at 65 536 blocks (`shape_kernel_durations.csv`)\n")
w("| shape, blocks / launch | path | kernel | rocprofv3 mean (median, min) | algorithmic bytes / launch | achieved | of 8 TB/s | round 2 (mean) |")
w("|---|---|---|---|---|---|---|---|")
for r in rows("shape_kernel_durations.csv"):
    K, M, L, B = (int(x) for x in r["shape_batch"].split("_"))
    byt = bytes_per_block(r["path"], K, M) * B
    gb = byt / (float(r["mean_us"]) * 1e-6) / 1e9
    r2 = R2S.get((r["shape_batch"], r["path"]))
    w("| K=%d M=%d L=%d, %d | %s | `%s` | %.1f us (%.1f, %.1f) | %s | %.2f TB/s | **%.1f %%** | %s |" % (K, M, L, B, LABEL[r["path"]], r["kernel"], float(r["mean_us"]), float(r["median_us"]),
      float(r["min_us"]), format(byt, ","), gb / 1e3, gb / 80, ("%.1f us = %.1f %%" % (r2, byt / (r2 * 1e-6) / 8e10)) if r2 else ""))
w("")
w("### HBM traffic per launch (`pmc_hbm_traffic_summary.csv`: FETCH_SIZE x 2 [gfx950 correction, calibrated in round 2 on copy kernels] + WRITE_SIZE)\n")
w("| run | kernel | fetched x 2 + written | algorithmic bytes | ratio |")
w("|---|---|---|---|---|")
acc = {}
for r in rows("pmc_hbm_traffic_summary.csv"):
    acc.setdefault((r["run"], r["kernel"]), {})[r["counter"]] = float(r["mean_KiB"])
for (run, k), v in sorted(acc.items()):
    if run == "probe" or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    parts = run.rsplit("_", 4)
    path, K, M, L, B = parts[0], int(parts[1]), int(parts[2]), int(parts[3]), int(parts[4])
    if ("modulate" in k) != (path == "modulate"):
        continue                                           # the input-preparation launches of the other runs
    tr = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    alg = bytes_per_block(path, K, M) * B
    w("| %s | `%s` | %s | %s | %.3f |" % (run, k, format(int(tr), ","), format(alg, ","), tr / alg))
w("")
w("### SQ counters per wave (`pmc_sq_counters_summary.csv`; counter / SQ_WAVES; the *_CYCLES / ACTIVE / WAIT counters tick once per four clocks)\n")
w("| run | kernel | wave life | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY | VALU instr. | LDS instr. | MFMA instr. | bank-conflict cycles | WAIT_INST_LDS |")
w("|---|---|---|---|---|---|---|---|---|---|---|")
sq = {}
for r in rows("pmc_sq_counters_summary.csv"):
    sq.setdefault((r["run"], r["kernel"]), {})[r["counter"]] = float(r["mean_KiB"])
for (run, k), v in sorted(sq.items()):
    if "SQ_WAVES" not in v or ("modulate" in k) != run.startswith("modulate"):
        continue
    n = v["SQ_WAVES"]
    g = lambda c: ("%.0f" % (v[c] / n)) if c in v else ""
    w("| %s | `%s` | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (run, k, g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY"), g("SQ_ACTIVE_INST_ANY"), g("SQ_INSTS_VALU"),
      g("SQ_INSTS_LDS"), g("SQ_INSTS_MFMA"), g("SQ_LDS_BANK_CONFLICT"), g("SQ_WAIT_INST_LDS")))
w("")
plain = json.load(open(os.path.join(D, "bench_default_plain.json")))
w("### bench.py on the same box (`bench_default_plain.json`; HIP event pair around a back-to-back run of 200 launches = kernel + dispatch gap)\n")
w("Headline `value` %.0f M blocks/s (200-step burst over %d streams; the demodulator works on frames modulated %d steps earlier, i.e. >= 256 MiB of traffic in between; demodulating the frames just written: %.0f M), sustained %.0f M over %.1f s, single stream %.0f M; `roofline` = %s: %.2f us, %.1f %%, copy ceiling %.0f GB/s." % (
    plain["value"] / 1e6, plain["config"]["streams"], plain["demod_lag_steps"], (plain["value_same_slot"] or 0) / 1e6, plain["sustained"]["value"] / 1e6, plain["sustained"]["seconds"],
    plain["value_single_stream"] / 1e6, plain["roofline"]["kernel"], plain["roofline"]["kernel_ms"] * 1e3, plain["roofline"]["frac"] * 100, plain["roofline"]["copy_ceiling_GBps"]))
w("`paths`: " + ", ".join("%s %.2f us (%.1f %%)" % (k, p["kernel_ms"] * 1e3, p["frac_of_hbm_peak"] * 100) for k, p in plain["paths"].items()) + ".")
w("`large_batch` (65 536 blocks): " + ", ".join("%s %.1f us back to back (%.1f %%), single launches %.1f-%.1f us (%.0f-%.0f %%), median %.1f us" % (
    k, p["kernel_ms"] * 1e3, p["frac_of_hbm_peak"] * 100, p["kernel_ms_per_launch_min"] * 1e3, p["kernel_ms_per_launch_max"] * 1e3, p["frac_of_hbm_peak_range"][0] * 100,
    p["frac_of_hbm_peak_range"][1] * 100, p["kernel_ms_per_launch_median"] * 1e3) for k, p in plain["large_batch"].items()) + ".")
cb = plain["cpu_baseline"]
w("`single_block_host_us` %.1f us (one `generic_work` with host pointers through `gfdm_python.Demodulator`); CPU port %.2f M blocks/s on %d pinned pthreads = %.1f x one thread (%s; %d CPUs in the affinity mask, cgroup CPU quota %s), %.3f M = %.1f us per block single thread.\n"
  % (plain["single_block_host_us"], cb["value"] / 1e6, cb["cores"], cb["scaling_vs_single_thread"], cb["cpu_model"], cb["cpus_in_affinity_mask"], cb["cgroup_cpu_quota"],
     cb["single_thread_value"] / 1e6, cb["single_thread_us_per_block"]))
w("| `bench.py --config` | metric | value (burst) | sustained | dominant kernel: ms, of 8 TB/s | CPU port (threads) |")
w("|---|---|---|---|---|---|")
for c in ("cfg3", "cfg4", "cfg5"):
    d = json.load(open(os.path.join(D, "bench_%s.json" % c)))
    w("| %s | %s | %.1f M blocks/s | %.1f M blocks/s | `%s` %.3f ms, %.1f %% | %.3f M blocks/s (%d) |" % (c, d["metric"], d["value"] / 1e6, d["sustained"]["value"] / 1e6,
      d["roofline"]["kernel"].split(" (")[0], d["roofline"]["kernel_ms"], d["roofline"]["frac"] * 100, d["cpu_baseline"]["value"] / 1e6, d["cpu_baseline"]["cores"]))
print("\n".join(out))
