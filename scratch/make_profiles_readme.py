"""profiles/<round>/* -> the measured tables of profiles/README.md (printed to stdout; the prose around them is in the template below)."""
import csv, json, sys, os
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
D = os.path.join("profiles", rnd)
N, K, M, B = 576, 64, 9, 4096


def rows(name):
    return list(csv.DictReader(open(os.path.join(D, name))))


def pick(rs, kernel, wgs, queue="1"):
    for r in rs:
        if r["kernel"] == kernel and int(r["workgroups"]) == wgs and r["queue"] == queue:
            return r
    return None


bench = rows("bench_default_kernel_durations_by_launch_shape.csv")
est = rows("bench_est_kernel_durations_by_launch_shape.csv")
plain = json.load(open(os.path.join(D, "bench_default_plain.json")))
under = json.load(open(os.path.join(D, "bench_default_under_rocprofv3.json")))

out = []
w = out.append
w("### Kernel durations, K=64 M=9 L=2, one kernel on the GPU at a time (default stream)\n")
w("From `bench_default_kernel_durations_by_launch_shape.csv` (rocprofv3 begin/end timestamps; template arguments: K, M, L, mode")
w("(0 frequency-domain output, 1 demodulate, 2 IC), equaliser (0 none, 1 vector, 2 estimated in the kernel), real symmetric IC kernel):\n")
w("| kernel | blocks / launch | launches | rocprofv3 mean (median, min) | algorithmic bytes / launch | achieved (mean) | of 8 TB/s |")
w("|---|---|---|---|---|---|---|")
table = [("k_row_receive<64, 9, 2, 1, 0, false>", "MF demod (dominant kernel of the headline step)", 16),
         ("k_row_receive<64, 9, 2, 1, 1, false>", "ZF demod", 24),
         ("k_row_receive<64, 9, 2, 2, 0, true>", "MF + 2 IC", 16),
         ("k_row_receive<64, 9, 2, 2, 1, true>", "ZF + 2 IC (**north-star path**, BASELINE configs[2])", 24),
         ("k_row_modulate<64, 9, 2, 0>", "modulate", 16)]
for wgs, blocks in ((1024, 4096), (16384, 65536)):
    for kname, label, bps in table:
        r = pick(bench, kname, wgs)
        if not r:
            continue
        byt = bps * N * blocks
        gb = byt / (float(r["mean_us"]) * 1e-6) / 1e9
        w("| `%s` %s | %d | %s | %.2f us (%.2f, %.2f) | %s | %.2f TB/s | %.1f %% |" % (kname, label, blocks, r["launches"], float(r["mean_us"]),
          float(r["median_us"]), float(r["min_us"]), format(byt, ","), gb / 1e3, gb / 80))
r = pick(bench, "k_row_receive<64, 9, 2, 2, 2, true>", 1024)
if r:
    byt = (8 * N + 16 * K + 8 * 52 * M) * B
    gb = byt / (float(r["mean_us"]) * 1e-6) / 1e9
    w("| `k_row_receive<64, 9, 2, 2, 2, true>` estimator + ZF + 2 IC + demapper (52 active) in one kernel | 4096 | %s | %.2f us (%.2f, %.2f) | %s | %.2f TB/s | %.1f %% |"
      % (r["launches"], float(r["mean_us"]), float(r["median_us"]), float(r["min_us"]), format(byt, ","), gb / 1e3, gb / 80))
w("")
w("bench.py's own figures from the same box (`bench_default_plain.json`; HIP event pair around a back-to-back run of 200 launches = kernel +")
w("dispatch gap): " + ", ".join("%s %.2f us (%.1f %%)" % (k, p["kernel_ms"] * 1e3, p["frac_of_hbm_peak"] * 100) for k, p in plain["paths"].items()) + ".")
w("`roofline` (MF demod): %.2f us, %.1f %%; rocprofv3 mean of that kernel above.  Headline `value`: %.0f M blocks/s over %d streams (%.0f M on one stream); under rocprofv3: %.0f M."
  % (plain["roofline"]["kernel_ms"] * 1e3, plain["roofline"]["frac"] * 100, plain["value"] / 1e6, plain["config"]["streams"], plain["value_single_stream"] / 1e6, under["value"] / 1e6))
w("65 536 blocks per launch (`large_batch`): " + ", ".join("%s %.1f us back to back (%.1f %%), %.1f us median of single launches" % (k, p["kernel_ms"] * 1e3, p["frac_of_hbm_peak"] * 100, p["kernel_ms_per_launch_median"] * 1e3)
                                                           for k, p in plain["large_batch"].items()) + ".")
cb = plain["cpu_baseline"]
w("CPU baseline on the same box: %.2f M blocks/s on %d threads (%s), %.3f M single thread.\n" % (cb["value"] / 1e6, cb["cores"], cb["cpu_model"], cb["single_thread_value"] / 1e6))
w("### Channel estimator kernels (`bench_est_kernel_durations_by_launch_shape.csv`, `bench_est.txt`)\n")
w("| kernel | frames / launch | rocprofv3 mean (median, min) | algorithmic bytes / launch | achieved | of 8 TB/s |")
w("|---|---|---|---|---|---|")
for kname, label, shape in (("k_row_estimate<64, 9>", "estimate_frame", (64, 9, 52)), ("k_row_estimate<128, 15>", "estimate_frame", (128, 15, 110)),
                            ("k_row_estimate<256, 31>", "estimate_frame", (256, 31, 220)),
                            ("k_row_receive<64, 9, 2, 2, 2, true>", "estimator + ZF + 2 IC + demapper", (64, 9, 52)),
                            ("k_row_receive<64, 9, 2, 2, 1, true>", "ZF + 2 IC + demapper, f_eq given", (64, 9, 52)),
                            ("k_row_receive<128, 15, 4, 2, 2, true>", "estimator + ZF + 2 IC + demapper", (128, 15, 110)),
                            ("k_row_receive<128, 15, 4, 2, 1, true>", "ZF + 2 IC + demapper, f_eq given", (128, 15, 110))):
    k, m, a = shape
    n = k * m
    for r in est:
        if r["kernel"] != kname:
            continue
        frames = int(r["workgroups"]) * int(r["workgroup_size"]) // k
        if "estimate" in kname:
            byt = (16 * k + 8 * n) * frames
        elif ", 2, true>" in kname:
            byt = (8 * n + 16 * k + 8 * a * m) * frames
        else:
            byt = (16 * n + 8 * a * m) * frames
        gb = byt / (float(r["mean_us"]) * 1e-6) / 1e9
        w("| `%s` %s | %d | %.2f us (%.2f, %.2f) | %s | %.2f TB/s | %.1f %% |" % (kname, label, frames, float(r["mean_us"]), float(r["median_us"]),
          float(r["min_us"]), format(byt, ","), gb / 1e3, gb / 80))
print("\n".join(out))
