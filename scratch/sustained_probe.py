"""Why does one kernel read differently back to back and launch by launch?  (VERDICT r04, weak item 4 / task 2b)

usage: sustained_probe.py <copy|modulate|demod_mf|demod_zf|demod_mf_ic2|demod_zf_ic2> <blocks> [K M L] [seconds]

For ONE kernel at ONE batch, in one process, on a ring of buffer sets larger than the Infinity Cache:
  burst      back-to-back runs of 10 launches inside one HIP event pair (what bench.py's large_batch `kernel_ms` was), 6 times, a
             synchronize in between; the first burst follows host-side set-up (GPU idle)
  pairs      10 launches, each inside its own queued event pair, 6 times
  sustained  launches back to back for `seconds`, an event every `chunk` launches: the series of per-launch means over the run
  pairs_sus  the same length with an event pair around every launch: median per tenth of the run
While the phases run, a thread samples the GPU's shader / memory clock and socket power from sysfs (whatever the box lets an ordinary
user read).  Everything is printed as text; scratch/gpu_r5.sh collects it under gpurun_out/.
"""
import glob
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
import numpy as np
import torch
import gfdm_amd
from gfdm_amd import synth
from gfdm_amd.filters import get_frequency_domain_filter

path, B = sys.argv[1], int(sys.argv[2])
K, M, L = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (64, 9, 2)
seconds = float(sys.argv[6]) if len(sys.argv) > 6 else 2.0
N = K * M
dev = torch.device("cuda:0")
bps = 24 if "zf" in path else 16
nbytes = bps * N * B


class Sampler(threading.Thread):
    """current sclk / mclk (MHz) and power (W) from sysfs, every ~2 ms"""

    def __init__(self):
        super().__init__(daemon=True)
        self.files = {}
        for card in sorted(glob.glob("/sys/class/drm/card*/device")):
            if os.path.exists(os.path.join(card, "pp_dpm_sclk")):
                self.files["sclk"] = os.path.join(card, "pp_dpm_sclk")
                self.files["mclk"] = os.path.join(card, "pp_dpm_mclk")
                self.files["fclk"] = os.path.join(card, "pp_dpm_fclk")
                for p in glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_input")):
                    self.files["power"] = p
                for p in glob.glob(os.path.join(card, "hwmon", "hwmon*", "freq1_input")):
                    self.files["freq1"] = p
                break
        self.rows = []
        self.stop = False

    @staticmethod
    def _dpm(text):
        for line in text.splitlines():
            if line.rstrip().endswith("*"):
                return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        return float("nan")

    def read(self):
        row = {"t": time.perf_counter()}
        for k, p in self.files.items():
            try:
                s = open(p).read()
                row[k] = self._dpm(s) if k in ("sclk", "mclk", "fclk") else float(s) / 1e6
            except (OSError, ValueError, IndexError):
                row[k] = float("nan")
        return row

    def run(self):
        while not self.stop:
            self.rows.append(self.read())
            time.sleep(0.002)

    def window(self, t0, t1):
        sel = [r for r in self.rows if t0 <= r["t"] <= t1]
        out = []
        for k in ("sclk", "freq1", "mclk", "fclk", "power"):
            v = np.array([r[k] for r in sel if k in r and r[k] == r[k]])
            if len(v):
                out.append("%s %.0f/%.0f/%.0f" % (k, v.min(), np.median(v), v.max()))
        return ("%d samples: " % len(sel)) + (", ".join(out) if out else "nothing readable") + "  (min/median/max; MHz, W)"


taps = get_frequency_domain_filter("rrc", 0.1 if K == 256 else 0.2, M, K, L)
mod = gfdm_amd.Modulator(M, K, L, taps)
dem = gfdm_amd.Demodulator(M, K, L, taps)
adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, np.arange(K), 2, np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2))
nbuf = 3 if "zf" in path else 2
slots = max(3, min(64, (2 << 30) // (nbuf * 8 * N * B)))
CH = 8192
data = []
for s in range(slots):
    sym = torch.empty(B, N, dtype=torch.complex64, device=dev)
    x = torch.empty_like(sym)
    f = torch.empty_like(sym) if "zf" in path else None
    for c in range(0, B, CH):
        n = min(CH, B - c)
        sym[c:c + n] = synth.qpsk_symbols(s * B + c, n, N, dev)
        x[c:c + n] = mod.modulate(sym[c:c + n])
        if f is not None:
            f[c:c + n] = synth.channel_response(s * B + c, n, N, dev)
            x[c:c + n] = synth.through_channel(x[c:c + n], f[c:c + n])
    data.append((sym if path == "modulate" else x, f, torch.empty_like(sym)))
    if path != "modulate":
        del sym
torch.cuda.synchronize()


def go(i):
    x, f, o = data[i % slots]
    if path == "copy":                                 # control: a plain device copy of the same 16 N bytes per block (torch's copy kernel)
        o.copy_(x)
    elif path == "modulate":
        mod.modulate(x, out=o)
    elif path == "demod_mf":
        dem.demodulate(x, out=o)
    elif path == "demod_zf":
        dem.demodulate_equalize(x, f, out=o)
    elif path == "demod_mf_ic2":
        adv.demodulate(x, out=o)
    elif path == "demod_zf_ic2":
        adv.demodulate_equalize(x, f, out=o)
    else:
        raise SystemExit("unknown path " + path)


# shader clock as the GPU runs it (scratch/probe/sclk_probe.hip): a one-wave kernel queued on the stream, 20 us of the 100 MHz counter each
import ctypes
_pl = ctypes.CDLL(os.path.join(ROOT, "scratch", "probe", "libsclk_probe.so"))
_pl.sclk_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong]
_probe_buf = torch.zeros(4096, 2, dtype=torch.int64, device=dev)
_probe_n = [0]


def probe(ticks=2000):
    """queue a clock probe of `ticks` x 10 ns; returns its index"""
    i = _probe_n[0]
    _probe_n[0] += 1
    assert _pl.sclk_probe(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(_probe_buf[i].data_ptr()), ticks) == 0
    return i


def mhz(i):
    c, r = (int(v) for v in _probe_buf[i].cpu())
    return 100.0 * c / max(r, 1)


ev = lambda: torch.cuda.Event(enable_timing=True)
frac = lambda ms: nbytes / (ms * 1e-3) / 8e12
smp = Sampler()
print("%s K=%d M=%d L=%d, %d blocks per launch, %.1f MB algorithmic bytes per launch, ring of %d buffer sets; build %s" % (path, K, M, L, B, nbytes / 1e6, slots, gfdm_amd.build_id()))
print("sysfs: " + (", ".join("%s=%s" % kv for kv in smp.files.items()) or "no pp_dpm_sclk readable"))
print("idle   " + smp.window(0, 0).split(":")[0] + " | one reading: " + str({k: v for k, v in smp.read().items() if k != "t"}))
smp.start()
for i in range(slots):
    go(i)
torch.cuda.synchronize()
time.sleep(0.5)                                    # GPU idle, as after bench.py's host-side set-up

n = 10
for rep in range(6):
    t0 = time.perf_counter()
    a, b = ev(), ev()
    p0 = probe()
    a.record()
    for i in range(n):
        go(rep * n + i)
    b.record()
    p1 = probe()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print("burst %d     %d launches back to back: mean %.4f ms  frac %.3f | shader clock in front %.0f MHz, behind %.0f MHz | %s"
          % (rep, n, ms, frac(ms), mhz(p0), mhz(p1), smp.window(t0, time.perf_counter())))
    time.sleep(0.05 * rep)                             # growing idle gaps between the bursts
for rep in range(6):
    t0 = time.perf_counter()
    pairs = [(ev(), ev()) for _ in range(n)]
    clk = []
    for i in range(n):
        pairs[i][0].record()
        go(rep * n + i)
        pairs[i][1].record()
        clk.append(probe(300))                         # 3 us of the shader clock right behind every launch
    torch.cuda.synchronize()
    ts = [x.elapsed_time(y) for x, y in pairs]
    t = sorted(ts)
    print("pairs %d     %d launches, a pair each: median %.4f ms (min %.4f max %.4f)  frac %.3f | %s" % (rep, n, t[n // 2], t[0], t[-1], frac(t[n // 2]), smp.window(t0, time.perf_counter())))
    print("            launch by launch, ms @ shader MHz behind it: " + "  ".join("%.4f@%.0f" % (a, mhz(c)) for a, c in zip(ts, clk)))

# sustained back to back
est = ms * 1e-3
chunk = max(1, int(0.02 / est))                    # an event every ~20 ms
nch = max(10, int(seconds / (chunk * est)))
marks = [ev() for _ in range(nch + 1)]
t0 = time.perf_counter()
marks[0].record()
clk = []
for c in range(nch):
    for i in range(chunk):
        go(c * chunk + i)
    marks[c + 1].record()
    if c % max(1, nch // 10) == 0:
        clk.append(probe())
torch.cuda.synchronize()
t1 = time.perf_counter()
series = np.array([marks[c].elapsed_time(marks[c + 1]) / chunk for c in range(nch)])
tot = marks[0].elapsed_time(marks[nch]) / (nch * chunk)
print("sustained   %d launches back to back in %.2f s: mean %.4f ms  frac %.3f | %s" % (nch * chunk, t1 - t0, tot, frac(tot), smp.window(t0, t1)))
print("            per-launch mean of each tenth of the run (ms): " + " ".join("%.4f" % series[i * nch // 10:(i + 1) * nch // 10].mean() for i in range(10)))
print("            shader clock through the run (MHz): " + " ".join("%.0f" % mhz(i) for i in clk))
print("            first five %d-launch chunks (ms): %s ; slowest chunk %.4f, fastest %.4f" % (chunk, " ".join("%.4f" % v for v in series[:5]), series.max(), series.min()))

# sustained, a pair around every launch
npl = min(nch * chunk, 20000)
pairs = [(ev(), ev()) for _ in range(npl)]
t0 = time.perf_counter()
for i in range(npl):
    pairs[i][0].record()
    go(i)
    pairs[i][1].record()
torch.cuda.synchronize()
t1 = time.perf_counter()
t = np.array([x.elapsed_time(y) for x, y in pairs])
span = pairs[0][0].elapsed_time(pairs[-1][1]) / npl
print("pairs_sus   %d launches, a pair each, in %.2f s: median %.4f ms  frac %.3f ; first event to last event / launches = %.4f ms  frac %.3f | %s"
      % (npl, t1 - t0, np.median(t), frac(float(np.median(t))), span, frac(span), smp.window(t0, t1)))
print("            median of each tenth of the run (ms): " + " ".join("%.4f" % np.median(t[i * npl // 10:(i + 1) * npl // 10]) for i in range(10)))
smp.stop = True
