"""Worst measured error per path from a GFDM_ERRLOG file (tests/conftest.py::check_err): python scratch/errlog_table.py <errlog>"""
import collections, re, sys
worst = collections.defaultdict(lambda: (0.0, 0.0, 0))
for line in open(sys.argv[1]):
    tag, err, tol = line.split()
    fam = re.sub(r"_(\d+_\d+_\d+.*|ic_.*|ref_.*|cfg.*|rxl_.*)$", "", tag)
    fam = re.sub(r"_\d+_\d+$", "", fam)
    fam = {"golden_ic_mf": "MF + IC vs pygfdm rounds", "golden_ic_zf": "ZF + IC vs pygfdm rounds", "golden": "MF / ZF + IC vs the pygfdm IC rounds",
           "golden_rx_overlap": "receiver at overlap 2 .. 8 vs pygfdm gfdm_demodulate_fft_loop", "golden_rx_overlap_S": "fft_filter_downsample at overlap 2 .. 8 vs the same model"}.get(fam, fam)
    w, t, n = worst[fam]
    worst[fam] = (max(w, float(err)), float(tol), n + 1)
print("| path | comparisons | worst relative error | bound |\n|---|---|---|---|")
for fam, (w, t, n) in sorted(worst.items()):
    print("| %s | %d | %.2e | %.0e |" % (fam, n, w, t))
