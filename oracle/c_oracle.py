"""ctypes binding for oracle/libgfdm_oracle.so (the plain-C CPU oracle).

TEST INFRASTRUCTURE ONLY -- see oracle/gfdm_oracle.c.  Used by tests/, smoke()
and bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

DECIDE = {"nearest": 0, "qpsk": 1, "bpsk": 2}


def build(cflags=None, out=None):
    """(Re)build the shared object; returns its path."""
    out = out or os.path.join(_HERE, "libgfdm_oracle.so")
    cmd = ["gcc"] + (cflags or ["-O3", "-march=x86-64-v3"]) + ["-fPIC", "-std=c11", "-shared", "-o", out,
                                                              os.path.join(_HERE, "gfdm_oracle.c"),
                                                              os.path.join(_HERE, "gfdm_oracle_bench.c"), "-lm", "-lpthread"]
    subprocess.check_call(cmd)
    return out


def load(path=None):
    global _LIB
    if path is None and _LIB is not None:
        return _LIB
    p = path or os.path.join(_HERE, "libgfdm_oracle.so")
    if not os.path.exists(p):
        build(out=p)
    lib = ctypes.CDLL(p)
    fp, ip, vp, lg = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int), ctypes.c_void_p, ctypes.c_long
    lib.gfdm_oracle_create.restype = vp
    lib.gfdm_oracle_create.argtypes = [ctypes.c_int] * 3 + [vp, ctypes.c_int]
    lib.gfdm_oracle_destroy.argtypes = [vp]
    lib.gfdm_oracle_block_size.argtypes = [vp]
    lib.gfdm_oracle_filter_taps.argtypes = [vp, vp]
    lib.gfdm_oracle_ic_filter_taps.argtypes = [vp, vp]
    lib.gfdm_oracle_modulate.argtypes = [vp, vp, vp, lg]
    lib.gfdm_oracle_fft_filter_downsample.argtypes = [vp, vp, vp, vp, lg]
    lib.gfdm_oracle_transform_subcarriers_to_td.argtypes = [vp, vp, vp, lg]
    lib.gfdm_oracle_cancel_sc_interference.argtypes = [vp, vp, vp, vp, lg]
    lib.gfdm_oracle_demodulate.argtypes = [vp, vp, vp, vp, lg]
    lib.gfdm_oracle_advanced_receive.argtypes = [vp, vp, vp, vp, lg, vp, ctypes.c_int, vp, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.gfdm_oracle_tx_create.restype = vp
    lib.gfdm_oracle_tx_create.argtypes = [ctypes.c_int] * 6 + [vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int, vp,
                                          ctypes.c_int, vp, ctypes.c_int]
    lib.gfdm_oracle_tx_destroy.argtypes = [vp]
    lib.gfdm_oracle_tx_input_vector_size.argtypes = [vp]
    lib.gfdm_oracle_tx_output_vector_size.argtypes = [vp]
    lib.gfdm_oracle_tx_work.argtypes = [vp, vp, vp, ctypes.c_int, lg, ctypes.c_int]
    lib.gfdm_oracle_bench.restype = lg
    lib.gfdm_oracle_bench.argtypes = [ctypes.c_int] * 3 + [vp] + [ctypes.c_int] * 5 + [vp, ctypes.c_double, ctypes.c_int, vp]
    del fp, ip
    if path is None:
        _LIB = lib
    return lib


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


class COracle:
    """One reference-style kernel object: single-threaded, one block per inner call."""

    def __init__(self, M, K, L, taps, lib=None):
        self.lib = lib or load()
        self.M, self.K, self.L, self.N = M, K, L, M * K
        t = _c64(taps)
        self.h = self.lib.gfdm_oracle_create(M, K, L, t.ctypes.data, t.size)
        if not self.h:
            raise ValueError("number of frequency taps MUST be equal to n_timeslots * overlap")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.gfdm_oracle_destroy(self.h)
            self.h = None

    def filter_taps(self):
        out = np.empty(self.M * self.L, np.complex64)
        self.lib.gfdm_oracle_filter_taps(self.h, out.ctypes.data)
        return out

    def ic_filter_taps(self):
        out = np.empty(self.M, np.complex64)
        self.lib.gfdm_oracle_ic_filter_taps(self.h, out.ctypes.data)
        return out

    def _run(self, fn, *arrays, extra=()):
        ins = [None if a is None else _c64(a) for a in arrays]
        first = ins[0]
        nblocks = first.size // self.N
        assert first.size == nblocks * self.N
        out = np.empty_like(first)
        ptrs = [None if a is None else a.ctypes.data for a in ins]
        fn(self.h, out.ctypes.data, *ptrs, nblocks, *extra)
        return out

    def modulate(self, x):
        return self._run(self.lib.gfdm_oracle_modulate, x)

    def fft_filter_downsample(self, x, f_eq=None):
        return self._run(self.lib.gfdm_oracle_fft_filter_downsample, x, f_eq)

    def transform_subcarriers_to_td(self, x):
        return self._run(self.lib.gfdm_oracle_transform_subcarriers_to_td, x)

    def cancel_sc_interference(self, td, fd):
        return self._run(self.lib.gfdm_oracle_cancel_sc_interference, td, fd)

    def demodulate(self, x, f_eq=None):
        return self._run(self.lib.gfdm_oracle_demodulate, x, f_eq)

    def advanced_receive(self, x, smap, points, ic_iter, f_eq=None, do_phase_compensation=0, kind="nearest"):
        smap = np.ascontiguousarray(smap, dtype=np.int32)
        pts = _c64(points)
        return self._run(self.lib.gfdm_oracle_advanced_receive, x, f_eq,
                         extra=(smap.ctypes.data, smap.size, pts.ctypes.data, pts.size, DECIDE[kind], ic_iter,
                                do_phase_compensation))


BENCH_MODE = {"mod_demod": 0, "demod": 1, "demod_ic": 2, "mod": 3}


def bench_threads(M, K, L, taps, mode, nthreads, seconds, use_eq=False, ic_iter=0, cpus=None, chunk=32, lib=None):
    """gfdm_oracle_bench: `nthreads` pinned pthreads, one kernel object each, block after block until the deadline.
    Returns (blocks processed by all threads, elapsed seconds)."""
    lib = lib or load()
    t = _c64(taps)
    cpu_arr = None if cpus is None else np.ascontiguousarray(cpus[:nthreads], dtype=np.int32)
    el = ctypes.c_double(0.0)
    n = lib.gfdm_oracle_bench(M, K, L, t.ctypes.data, t.size, BENCH_MODE[mode], int(bool(use_eq)), int(ic_iter), int(nthreads),
                              None if cpu_arr is None else cpu_arr.ctypes.data, float(seconds), int(chunk), ctypes.byref(el))
    if n < 0:
        raise RuntimeError("gfdm_oracle_bench failed")
    return int(n), el.value


class COracleTx:
    """Composite transmitter oracle (mapper -> modulator -> cyclic prefix + ramp -> preamble), one port per call."""

    def __init__(self, M, K, A, cp, cs, ramp, smap, per_timeslot, L, taps, window, shifts, preambles, lib=None):
        self.lib = lib or load()
        t = _c64(taps); w = _c64(window); p = _c64(np.atleast_2d(preambles))
        sm = np.ascontiguousarray(smap, np.int32); sh = np.ascontiguousarray(shifts, np.int32)
        self.h = self.lib.gfdm_oracle_tx_create(M, K, A, cp, cs, ramp, sm.ctypes.data, int(per_timeslot), L, t.ctypes.data, t.size,
                                                w.ctypes.data, w.size, sh.ctypes.data, sh.size, p.ctypes.data, p.shape[1])
        if not self.h:
            raise ValueError("invalid transmitter configuration")
        self.n_in = self.lib.gfdm_oracle_tx_input_vector_size(self.h)
        self.n_out = self.lib.gfdm_oracle_tx_output_vector_size(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.gfdm_oracle_tx_destroy(self.h)
            self.h = None

    def work(self, symbols, port=0):
        x = _c64(np.atleast_2d(symbols))
        out = np.empty((x.shape[0], self.n_out), np.complex64)
        self.lib.gfdm_oracle_tx_work(self.h, out.ctypes.data, x.ctypes.data, x.shape[1], x.shape[0], port)
        return out
