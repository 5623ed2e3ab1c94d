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
x = c2("128_15_4", "demod_mf_ic2"); t2.append("| K=128 M=15 L=4 | **MF + 2 IC (configs[3])**, rounds on the matrix cores (default) | %s; other collections of the round: 60.7 (58.9), 61.4 (57.9), 62.4 (63.0) | %s; 450.0, 452.2, 455.5 | 62.7 us = 50.2 %% / 466.6 us = 53.9 %% |" % x)
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
t.append('   | MF demodulate, buffers registered once (used in place) | %.3f | %.2f | %.1f | %.1f | %.1f (88 GB/s over the link, both directions) |' % tuple(f('host_batch_demod_mf', 'registered', n) for n in ('1', '16', '256', '4096', '65536')))
t.append('   | ZF + 2 IC (three pointers per block), pageable / registered | %s |' % ' | '.join('%.2g / %.2g' % (f('host_batch_zf_ic2', 'pageable', n), f('host_batch_zf_ic2', 'registered', n)) for n in ('1', '16', '256', '4096', '65536')))
s = s[:a] + '\n'.join(t) + '\n\n' + s[b:]
open(p, 'w').write(s)
print("tables refreshed from", D)
