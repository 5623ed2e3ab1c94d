"""Rewrites the measured tables of DESIGN.md (section 6: K=64 paths, configs[3] / [4]; section 16: host path) and the INTEGRATION.md host table from profiles/r04/*."""
import csv, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "profiles", "r04")
rows = {r['label']: r for r in csv.DictReader(open(os.path.join(D, 'kernel_alone.csv')))}
BPS = {"modulate": 16, "demod_mf": 16, "demod_mf_ic2": 16, "demod_zf": 24, "demod_zf_ic2": 24}
def cell(label):
    r = rows[label]; m = re.match(r"(\d+)_(\d+)_(\d+)_(.+)_(\d+)$", label); K, M, L, path, B = int(m[1]), int(m[2]), int(m[3]), m[4], int(m[5])
    byt = BPS[path] * K * M * B; mean = float(r['mean_us']); med = float(r['median_us'])
    return mean, med, byt / (mean * 1e-6) / 1e12, 100 * byt / (mean * 1e-6) / 8e12, 100 * byt / (med * 1e-6) / 8e12
R3 = {"demod_mf": ("10.70 us (44.1 %); 9.88 us on box B", "106.5 us (70.9 %)"), "demod_zf": ("13.36 us (53.0 %)", "147.2 us (76.9 %)"), "demod_zf_ic2": ("15.08 us (46.9 %)", "166.8 us (67.9 %)"),
      "demod_mf_ic2": ("12.72 us (37.1 %)", "135.3 us (55.8 %)"), "modulate": ("11.42 us (41.3 %)", "107.6 us (70.2 %)")}
NAME = {"demod_mf": "demod MF", "demod_zf": "demod ZF", "demod_zf_ic2": "demod ZF + 2 IC (**north star**, target >= 40 %)", "demod_mf_ic2": "demod MF + 2 IC", "modulate": "modulate"}
t1 = ["| path (K=64, M=9, L=2) | 4096 blocks / launch: mean (median) | round 3 | 65 536 blocks / launch | round 3 |", "|---|---|---|---|---|"]
for p in ("demod_mf", "demod_zf", "demod_zf_ic2", "demod_mf_ic2", "modulate"):
    a = cell("64_9_2_%s_4096" % p); b = cell("64_9_2_%s_65536" % p)
    t1.append("| %s | %.2f us (%.2f) = %.2f TB/s (**%.1f %%**) | %s | %.1f us = %.2f TB/s (**%.1f %%**) | %s |" % (NAME[p], a[0], a[1], a[2], a[3], R3[p][0], b[0], b[2], b[3], R3[p][1]))
t2 = ["| shape | path | 8192 blocks: mean (median) | 65 536 blocks | round 3 (8192 / 65 536) |", "|---|---|---|---|---|"]
def c2(shape, p):
    a = cell("%s_%s_8192" % (shape, p)); b = cell("%s_%s_65536" % (shape, p))
    return "%.1f us (%.1f) = **%.1f %%**" % (a[0], a[1], a[3]), "%.1f us = **%.1f %%**" % (b[0], b[3])
x = c2("128_15_4", "demod_mf"); t2.append("| K=128 M=15 L=4 | MF demod | %s | %s | 50.7 / 377 us |" % x)
x = c2("128_15_4", "demod_mf_ic2"); t2.append("| K=128 M=15 L=4 | **MF + 2 IC (configs[3])**, rounds on the matrix cores (default) | %s; another box 60.0 (`ic_wave_local_store_ab.txt`), same-box A/B of the tile padding 61.9 -> 59.3; before the padding, four collections: 60.7-62.4 | %s; another box 445; A/B 456 -> 445; before: 450-456 | 62.7 us = 50.2 %% / 466.6 us = 53.9 %% |" % x)
t2.append("| K=128 M=15 L=4 | the same with the rounds on the vector ALU (`set_ic_matrix_cores(0)`; kernel unchanged since round 3) | 66.7-72.7 us over the boxes = 43-47 % | 518-521 us = 48 % | 66.7 / 518.6 us |")
x = c2("128_15_4", "demod_zf_ic2"); t2.append("| K=128 M=15 L=4 | ZF + 2 IC, matrix cores | %s | %s | 73.6 us = 64.1 %% / 564 us = 66.9 %% |" % x)
a = c2("256_31_2", "demod_mf"); b = c2("256_31_2", "modulate")
t2.append("| K=256 M=31 L=2 | MF demod / modulate | %s / %s | %s / %s | 278.9, 273.2 / 2063, 2056 us |" % (a[0].replace('**', ''), b[0].replace('**', ''), a[1].replace('**', ''), b[1].replace('**', '')))
x = c2("256_31_2", "demod_zf"); t2.append("| K=256 M=31 L=2 | **ZF demod (configs[4])** | %s | %s | 348.5 us = 56.0 %% / 2560 us = 61.0 %% |" % x)
p = os.path.join(ROOT, 'DESIGN.md')
s = open(p).read()
a = s.index('| path (K=64, M=9, L=2) | 4096 blocks / launch: mean (median) | round 3 |'); b = s.index('`bench.py` reports, per kernel, the MEDIAN OF HIP EVENT PAIRS')
s = s[:a] + '\n'.join(t1) + '\n\n' + s[b:]
a = s.index('| shape | path | 8192 blocks: mean (median) | 65 536 blocks | round 3 (8192 / 65 536) |'); b = s.index('The review of round 3 set MF + 2 IC of configs[3] <= 58 us')
s = s[:a] + '\n'.join(t2) + '\n\n' + s[b:]
r = json.load(open(os.path.join(D, 'bench_default.json')))
P = r['paths']
def hrow(k, kind):
    p_ = P[k][kind]; out = []
    for n in ('1', '16', '256', '4096', '65536'):
        v = p_[n]['blocks_per_s'] / 1e6
        if n == '1': out.append('%.3f (%.0f us)' % (v, p_[n]['us_per_call']))
        elif n in ('4096', '65536'): out.append(('**%.1f**' if n == '65536' else '%.1f') % v + ' (%.0f GB/s)' % p_[n]['link_GBps'])
        else: out.append('%.2f' % v if v < 1 else '%.1f' % v)
    return out
a = s.index('| path | memory | 1 block / call | 16 | 256 | 4096 | 65 536 | CPU port of the reference algorithm, 1 thread / 16 threads |'); b = s.index("Read against the reference's own deployment")
t = ['| path | memory | 1 block / call | 16 | 256 | 4096 | 65 536 | CPU port of the reference algorithm, 1 thread / 16 threads |', '|---|---|---|---|---|---|---|---|']
for k, name in (('host_batch_modulate', 'modulate'), ('host_batch_demod_mf', 'MF demodulate'), ('host_batch_zf_ic2', 'ZF + 2 IC')):
    cpu = P[k]['cpu_port']
    t.append('| %s | pageable | %s | %.3f / %.2f |' % (name, ' | '.join(hrow(k, 'pageable')), cpu['single_thread_blocks_per_s'] / 1e6, cpu['blocks_per_s'] / 1e6))
    t.append('| %s | registered | %s | |' % (name, ' | '.join(hrow(k, 'registered'))))
s = s[:a] + '\n'.join(t) + '\n\n' + s[b:]
open(p, 'w').write(s)
p = os.path.join(ROOT, 'INTEGRATION.md')
s = open(p).read()
a = s.index('   | blocks per `work()` call | 1 | 16 | 256 | 4096 | 65 536 |'); b = s.index("3. The wrappers' `work()` bodies become one call into")
f = lambda k, kind, n: P[k][kind][n]['blocks_per_s'] / 1e6
t = ['   | blocks per `work()` call | 1 | 16 | 256 | 4096 | 65 536 |', '   |---|---|---|---|---|---|']
t.append('   | MF demodulate, pageable buffers (bounced in chunks) | %.3f | %.2f | %.1f | %.1f | %.1f |' % tuple(f('host_batch_demod_mf', 'pageable', n) for n in ('1', '16', '256', '4096', '65536')))
t.append('   | MF demodulate, buffers registered once (used in place) | %.3f | %.2f | %.1f | %.1f | %.1f (%.0f GB/s over the link, both directions) |' % (tuple(f('host_batch_demod_mf', 'registered', n) for n in ('1', '16', '256', '4096', '65536')) + (P['host_batch_demod_mf']['registered']['65536']['link_GBps'],)))
t.append('   | ZF + 2 IC (three pointers per block), pageable / registered | %s |' % ' | '.join('%.2g / %.2g' % (f('host_batch_zf_ic2', 'pageable', n), f('host_batch_zf_ic2', 'registered', n)) for n in ('1', '16', '256', '4096', '65536')))
s = s[:a] + '\n'.join(t) + '\n\n' + s[b:]
open(p, 'w').write(s)
print("tables refreshed from", D)

# ---- narrative numbers: headline paragraph and SQ table of DESIGN.md, host paragraph of README.md ----
c3 = json.load(open(os.path.join(D, 'bench_cfg3.json'))); c4 = json.load(open(os.path.join(D, 'bench_cfg4.json'))); c5 = json.load(open(os.path.join(D, 'bench_cfg5.json')))
p = os.path.join(ROOT, 'DESIGN.md')
s = open(p).read()
a = s.index('Headline (`bench.py`, configs[1], step = modulate + MF demodulate of 4096 blocks, `profiles/r04/bench_default.json`)')
b = s.index('## 7. Where the time goes')
new = '''Headline (`bench.py`, configs[1], step = modulate + MF demodulate of 4096 blocks, `profiles/r04/bench_default.json`): **%.0f M blocks/s** in the 200-step burst with the
independent steps pipelined over %d HIP streams (three or four sustain the same rate, three start a burst 3 %% faster: `profiles/r04/bench_streams_sweep.txt`) -- the demodulator
of a step works on the frames modulated %d steps EARLIER (%.0f MB of other traffic in between, more than the 256 MiB Infinity Cache; every step is still one modulate + one demodulate of a whole batch); demodulating the frames the same step has just written gives %.0f M, i.e. the
cache makes no measurable difference -- , **%.0f M blocks/s = %.0f GSym/s sustained over 1.9 s**, %.0f M on one stream (the boxes of the pool: 278-293 / 292-307 M).  CPU beside it
(same box, plain-C port of the reference algorithm, one kernel object per pinned pthread, set-up not timed, `oracle/gfdm_oracle_bench.c`): **%.2f M blocks/s on 16 threads** -- the
process sees 256 logical CPUs of 2 x EPYC 9575F in its affinity mask but its cgroup grants 16 CPUs' worth of run time (`cpu.max`); `bench.py` reads the quota, runs that many threads
and reports `cores`, `cgroup_cpu_quota`, `scaling_vs_single_thread`.  The other configurations (`bench.py --config`, burst / sustained): cfg3 %.0f / %.0f M blocks/s -- the north-star
path with independent batches pipelined over the same number of streams moves %.0f M x 13 824 B = %.1f TB/s of algorithmic bytes, **%.0f %% of the HBM peak** (a single 4096-block launch alone on the
GPU: 46-51 %%) --, cfg4 %.0f / %.0f M blocks/s (65 536 blocks per step on ONE GPU; dominant kernel %.0f us = %.1f %%), cfg5 %.1f / %.1f M blocks/s (%.0f us = %.1f %%).
`bench.py`'s headline hands device-resident buffers to the kernels; the host-buffer entry points are measured beside it (`paths.host_batch_*`, section 16).

''' % (r['value'] / 1e6, r['config']['streams'], r['demod_lag_steps'], r['demod_lag_steps'] * 32 * 576 * 4096 / 1e6, r['value_same_slot'] / 1e6, r['sustained']['value'] / 1e6, r['sustained']['value'] * 576 / 1e9, r['value_single_stream'] / 1e6, r['cpu_baseline']['value'] / 1e6,
       c3['value'] / 1e6, c3['sustained']['value'] / 1e6, c3['sustained']['value'] / 1e6, c3['sustained']['value'] * 13824 / 1e12, 100 * c3['sustained']['value'] * 13824 / 8e12,
       c4['value'] / 1e6, c4['sustained']['value'] / 1e6, c4['roofline']['kernel_ms'] * 1e3, 100 * c4['roofline']['frac'], c5['value'] / 1e6, c5['sustained']['value'] / 1e6,
       c5['roofline']['kernel_ms'] * 1e3, 100 * c5['roofline']['frac'])
s = s[:a] + new + s[b:]
sq = {}
for row in csv.DictReader(open(os.path.join(D, 'pmc_sq_counters_summary.csv'))):
    if not row['kernel'].startswith(('k_row_receive', 'k_row_modulate')): continue
    if row['kernel'].startswith('k_row_modulate') != row['run'].startswith('modulate'): continue
    sq.setdefault(row['run'], {})[row['counter']] = float(row['mean_KiB'])
g = lambda run, c: '%.0f' % (sq[run][c] / sq[run]['SQ_WAVES'])
names = [('demod_mf_64_9_2_4096', 'MF demod K=64 M=9, 4096', None), ('modulate_64_9_2_4096', 'modulate K=64 M=9, 4096', None), ('demod_mf_ic2_64_9_2_4096', 'MF + 2 IC K=64 M=9, 4096 (DPP rounds)', None),
         ('demod_zf_ic2_64_9_2_4096', 'ZF + 2 IC K=64 M=9, 4096', None), ('demod_zf_ic2_64_9_2_65536', 'ZF + 2 IC K=64 M=9, 65 536', None), ('demod_mf_128_15_4_8192', 'MF demod K=128 M=15 L=4, 8192', None),
         ('demod_mf_ic2_128_15_4_8192', 'MF + 2 IC K=128 M=15 L=4, 8192, rounds on the matrix cores',
          {'SQ_WAVE_CYCLES': 6880, 'SQ_WAIT_ANY': 2001, 'SQ_WAIT_INST_ANY': 2763, 'SQ_ACTIVE_INST_ANY': 2116, 'SQ_INSTS_VALU': 1402, 'SQ_INSTS_LDS': 177, 'SQ_LDS_BANK_CONFLICT': 383, 'SQ_WAIT_INST_LDS': 877}),
         ('demod_mf_ic2_128_15_4_8192_valu', 'the same, rounds on the vector ALU', None), ('demod_zf_256_31_2_8192', 'ZF demod K=256 M=31, 8192', None)]
cols = ['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_MFMA', 'SQ_LDS_BANK_CONFLICT', 'SQ_WAIT_INST_LDS']
t = ['| kernel, blocks per launch | wave lifetime | waiting for memory / barrier (`WAIT_ANY`) | issue stalls (`WAIT_INST_ANY`) | issuing (`ACTIVE_INST_ANY`) | VALU instr. | LDS instr. | MFMA | bank-conflict cycles | `WAIT_INST_LDS` |',
     '|---|---|---|---|---|---|---|---|---|---|']
for run, label, r3 in names:
    cells = []
    for c in cols:
        v = g(run, c)
        if r3 and c in r3: v = ('**%s (%d)**' if c in ('SQ_INSTS_LDS', 'SQ_WAIT_INST_LDS') else '%s (%d)') % (v, r3[c])
        cells.append(v)
    t.append('| %s | %s |' % (label, ' | '.join(cells)))
a = s.index('| kernel, blocks per launch | wave lifetime | waiting for memory / barrier (`WAIT_ANY`)'); b = s.index('(The shape-by-shape reading of these counters')
s = s[:a] + '\n'.join(t) + '\n\n' + s[b:]
open(p, 'w').write(s)
p = os.path.join(ROOT, 'README.md')
s = open(p).read()
a = s.index('**With host buffers**'); b = s.index('Shapes outside the compiled list are instantiated at run time')
new = '''**With host buffers** -- what gr-gfdm's GNU Radio wrappers hand over -- the `*_host` entry points run a chunked pipeline (DESIGN.md §16): pageable buffers are bounced through
pinned staging sets while the kernels work across the PCIe link, buffers registered once with `gfdm_hip_register_host` are used in place.  K=64 M=9 MF demodulation:
**%.1f M blocks/s** from pageable memory and **%.1f M blocks/s** (%.0f GB/s over the link, both directions) from registered memory at 65 536 blocks per call, %.1f / %.1f M at 4096,
%.2f / %.2f M at 16; ZF + 2 IC %.1f / %.1f M -- against %.2f M blocks/s for the plain-C port of the reference algorithm on one thread (how a GNU Radio block runs it) and %.1f M
on the 16 CPUs the box's cgroup grants (the pool's boxes: 5.3-6.7 M pageable).  One block per call costs 13-16 us (launch + completion latency: 3.8 us in the launch, ~8 us until
the GPU reports back), more than the CPU's 4.2 us: the GPU pays off through the batched `work()` bodies of `gfdm/batched_work.h`.

''' % (f('host_batch_demod_mf', 'pageable', '65536'), f('host_batch_demod_mf', 'registered', '65536'), P['host_batch_demod_mf']['registered']['65536']['link_GBps'], f('host_batch_demod_mf', 'pageable', '4096'), f('host_batch_demod_mf', 'registered', '4096'),
       f('host_batch_demod_mf', 'pageable', '16'), f('host_batch_demod_mf', 'registered', '16'), f('host_batch_zf_ic2', 'pageable', '65536'), f('host_batch_zf_ic2', 'registered', '65536'),
       P['host_batch_demod_mf']['cpu_port']['single_thread_blocks_per_s'] / 1e6, P['host_batch_demod_mf']['cpu_port']['blocks_per_s'] / 1e6)
s = s[:a] + new + s[b:]
open(p, 'w').write(s)
print("narrative numbers refreshed")

# ---- dominant kernels against torch's copy of the same bytes
rows = ["| `bench.py --config` | dominant kernel, blocks per launch | achieved | copy of the same bytes | kernel / copy | `frac` of 8 TB/s | HBM traffic / algorithmic bytes (PMC) |", "|---|---|---|---|---|---|---|"]
for f, name in (("bench_default", "default (configs[1]) | modulate K=64 M=9, 4096"), ("bench_cfg3", "cfg3 (configs[2], north star) | ZF + 2 IC K=64 M=9, 4096"),
                ("bench_cfg4", "cfg4 (configs[3]) | MF + 2 IC K=128 M=15 L=4, 65 536"), ("bench_cfg5", "cfg5 (configs[4]) | ZF K=256 M=31, 65 536")):
    rr = json.load(open(os.path.join(D, f + '.json')))['roofline']
    tr = ("%.3f" % (rr['traffic'] / rr['bytes_per_launch'])) if rr.get('traffic') else "n/a"
    rows.append("| %s | %.2f TB/s | %.2f TB/s | %.2f | %.1f %% | %s |" % (name, rr['achieved'] / 1e3, rr['copy_ceiling_GBps'] / 1e3, rr['achieved'] / rr['copy_ceiling_GBps'], 100 * rr['frac'], tr))
p2 = os.path.join(ROOT, 'DESIGN.md')
s2 = open(p2).read()
a = s2.index('| `bench.py --config` | dominant kernel, blocks per launch | achieved |'); b = s2.index("(`torch`'s copy is a reference point")
s2 = s2[:a] + '\n'.join(rows) + '\n\n' + s2[b:]
open(p2, 'w').write(s2)
print("copy table refreshed")
