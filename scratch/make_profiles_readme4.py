"""profiles/r04/* -> the measured tables of the r04 section of profiles/README.md (stdout)."""
import csv, json, os, re, sys
D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04")
rows = lambda n: list(csv.DictReader(open(os.path.join(D, n))))
out = []
w = out.append
LABEL = {"modulate": "modulate", "demod_mf": "MF demod", "demod_zf": "ZF demod", "demod_mf_ic2": "MF + 2 IC", "demod_zf_ic2": "ZF + 2 IC"}
BPS = {"modulate": 16, "demod_mf": 16, "demod_mf_ic2": 16, "demod_zf": 24, "demod_zf_ic2": 24}
# round-3 means of the same rows (profiles/r03, box A)
R3 = {"64_9_2_modulate_4096": 11.42, "64_9_2_demod_mf_4096": 10.70, "64_9_2_demod_zf_4096": 13.36, "64_9_2_demod_mf_ic2_4096": 12.72, "64_9_2_demod_zf_ic2_4096": 15.08,
      "64_9_2_modulate_65536": 107.6, "64_9_2_demod_mf_65536": 106.5, "64_9_2_demod_zf_65536": 147.2, "64_9_2_demod_mf_ic2_65536": 135.3, "64_9_2_demod_zf_ic2_65536": 166.8,
      "128_15_4_demod_mf_8192": 50.7, "128_15_4_demod_mf_ic2_8192": 62.7, "128_15_4_demod_zf_ic2_8192": 73.6, "128_15_4_demod_mf_65536": 377.0, "128_15_4_demod_mf_ic2_65536": 466.6,
      "128_15_4_demod_zf_ic2_65536": 564.0, "256_31_2_demod_mf_8192": 278.9, "256_31_2_modulate_8192": 273.2, "256_31_2_demod_zf_8192": 348.5, "256_31_2_demod_mf_65536": 2063.0,
      "256_31_2_modulate_65536": 2056.0, "256_31_2_demod_zf_65536": 2560.0}
build = open(os.path.join(D, "build_id.txt")).read().strip()
w("Library build `%s` (`gfdm_hip_build_id()`); every CSV of this directory carries it where the tool records it.\n" % build)
w("### Kernel durations, ONE kernel on the GPU at a time (`kernel_alone.csv`, `scratch/gpu_r4.sh alone`)\n")
w("`rocprofv3 --kernel-trace -- python3 scratch/run_kernel.py <path> <blocks> <launches> <ring slots> <K> <M> <L>`; template arguments of `k_row_receive`: K, M, L, mode")
w("(1 demodulate, 2 IC), equaliser (0 none, 1 vector), IC rounds (1 vector ALU with the real even kernel, 2 matrix cores).  Round-3 column: `profiles/r03`, another box.\n")
w("| shape | path | kernel | blocks / launch | rocprofv3 mean (median, min) us | algorithmic bytes / launch | achieved (mean) | of 8 TB/s (mean; median) | round 3 mean |")
w("|---|---|---|---|---|---|---|---|---|")
for r in rows("kernel_alone.csv"):
    m = re.match(r"(\d+)_(\d+)_(\d+)_(.+)_(\d+)$", r["label"])
    K, M, L, path, B = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4), int(m.group(5))
    byt = BPS[path] * K * M * B
    mean, med, mn = float(r["mean_us"]), float(r["median_us"]), float(r["min_us"])
    r3 = R3.get(r["label"])
    w("| K=%d M=%d L=%d | %s | `%s` | %d | %.2f (%.2f, %.2f) | %s | %.2f TB/s | **%.1f %%**; %.1f %% | %s |" % (K, M, L, LABEL[path], r["kernel"], B, mean, med, mn, format(byt, ","),
      byt / (mean * 1e-6) / 1e12, byt / (mean * 1e-6) / 8e10, byt / (med * 1e-6) / 8e10, ("%.1f us = %.1f %%" % (r3, byt / (r3 * 1e-6) / 8e10)) if r3 else ""))
w("")
w("### HBM traffic per launch (`pmc_hbm_traffic_summary.csv`: FETCH_SIZE x 2 [gfx950 correction, calibrated in round 2 on copy kernels] + WRITE_SIZE, separate `--pmc` passes)\n")
w("| run | kernel | fetched x 2 + written | algorithmic bytes | ratio |")
w("|---|---|---|---|---|")
acc = {}
for r in rows("pmc_hbm_traffic_summary.csv"):
    if not r["kernel"].startswith(("k_row_receive", "k_row_modulate")):
        continue
    acc.setdefault((r["run"], r["kernel"]), {})[r["counter"]] = float(r["mean_KiB"])
for (run, k), v in sorted(acc.items()):
    m = re.match(r"(.+)_(\d+)_(\d+)_(\d+)_(\d+)$", run)
    path, K, M, L, B = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    is_mod = k.startswith("k_row_modulate")
    if is_mod != (path == "modulate"):
        continue                                      # the modulator launches that prepare the receivers' inputs
    byt = BPS[path] * K * M * B
    tr = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    w("| %s | `%s` | %s | %s | %.4f |" % (run, k, format(int(tr), ","), format(byt, ","), tr / byt))
w("")
w("### SQ counters per wave (`pmc_sq_counters_summary.csv`; counter / SQ_WAVES; the *_CYCLES / ACTIVE / WAIT counters tick once per four clocks)\n")
w("| kernel, blocks per launch | wave lifetime | `WAIT_ANY` | `WAIT_INST_ANY` | `ACTIVE_INST_ANY` | VALU instr. | LDS instr. | MFMA | bank-conflict cycles | `WAIT_INST_LDS` |")
w("|---|---|---|---|---|---|---|---|---|---|")
sq = {}
for r in rows("pmc_sq_counters_summary.csv"):
    if not r["kernel"].startswith(("k_row_receive", "k_row_modulate")):
        continue
    if r["kernel"].startswith("k_row_modulate") != r["run"].startswith("modulate"):
        continue
    sq.setdefault((r["run"], r["kernel"]), {})[r["counter"]] = float(r["mean_KiB"])
for (run, k), v in sorted(sq.items()):
    wv = v.get("SQ_WAVES")
    if not wv:
        continue
    g = lambda c: ("%.0f" % (v[c] / wv)) if c in v else "-"
    w("| %s `%s` | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (run, k, g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY"), g("SQ_ACTIVE_INST_ANY"), g("SQ_INSTS_VALU"),
      g("SQ_INSTS_LDS"), g("SQ_INSTS_MFMA"), g("SQ_LDS_BANK_CONFLICT"), g("SQ_WAIT_INST_LDS")))
w("")
b = json.load(open(os.path.join(D, "bench_default.json")))
w("### The host-buffer batch path (`bench_default.json` -> `paths.host_batch_*`; K=64 M=9, `*_host` entry points, blocks per call)\n")
w("M blocks/s (GB/s of algorithmic bytes over the PCIe link, both directions together; us per call); `cpu_port` = the plain-C port of the reference algorithm on the same work.\n")
w("| path | memory | 1 | 16 | 256 | 4096 | 65 536 | CPU port, 1 thread | CPU port, %d threads |" % b["paths"]["host_batch_demod_mf"]["cpu_port"]["threads"])
w("|---|---|---|---|---|---|---|---|---|")
for name in ("host_batch_modulate", "host_batch_demod_mf", "host_batch_zf_ic2"):
    p = b["paths"][name]
    for kind in ("pageable", "registered"):
        cells = ["%.3f (%.1f; %.0f us)" % (p[kind][n]["blocks_per_s"] / 1e6, p[kind][n]["link_GBps"], p[kind][n]["us_per_call"]) for n in ("1", "16", "256", "4096", "65536")]
        cpu = p["cpu_port"]
        w("| %s | %s | %s | %s | %s |" % (name.replace("host_batch_", ""), kind, " | ".join(cells), ("%.3f" % (cpu["single_thread_blocks_per_s"] / 1e6)) if kind == "pageable" else "",
                                      ("%.3f" % (cpu["blocks_per_s"] / 1e6)) if kind == "pageable" else ""))
w("")
w("### bench.py, one line per configuration (`bench_default.json`, `bench_cfg3.json`, `bench_cfg4.json`, `bench_cfg5.json`)\n")
w("| config | value (M blocks/s) | sustained | dominant kernel: median of per-launch event pairs -> of 8 TB/s | pipelined (back-to-back mean) | traffic / algorithmic |")
w("|---|---|---|---|---|---|")
for c in ("default", "cfg3", "cfg4", "cfg5"):
    j = json.load(open(os.path.join(D, "bench_%s.json" % c)))
    rf = j["roofline"]
    w("| %s | %.1f | %s | `%s`: %.2f us -> **%.1f %%** | %.2f us -> %.1f %% | %s |" % (j["config"]["name"], j["value"] / 1e6, ("%.1f" % (j["sustained"]["value"] / 1e6)) if j.get("sustained") else "-",
      rf["kernel"].split(" (")[0], rf["kernel_ms"] * 1e3, 100 * rf["frac"], rf["kernel_ms_pipelined"] * 1e3, 100 * rf["frac_pipelined"],
      ("%.4f" % (rf["traffic"] / rf["bytes_per_launch"])) if rf.get("traffic") else "null (no PMC row for this kernel and batch)"))
print("\n".join(out))
