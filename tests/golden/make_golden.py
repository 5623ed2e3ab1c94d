#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference's own
Python model (`pygfdm`, /root/reference/python/pygfdm).

Runs ONLY in the build container (the reference checkout does not exist on the
GPU box).  The outputs (*.npz) are data: seeded inputs and the reference
model's outputs.  No reference source or bytecode is written anywhere
(sys.dont_write_bytecode).

What the reference computes here, and how its own tests use it
(python/qa_python_bindings.py:254-440, python/qa_simple_modulator_cc.py:39-97,
python/qa_simple_receiver_cc.py:38-83):
    ref_mod   = gfdm_modulate_block(get_data_matrix(d, K, False), taps, M, K, L, False)
    ref_demod = gfdm_demodulate_block(frame, taps, K, M, L)          (overlap 2 only)

Import notes (documented in DESIGN.md "Oracle"):
  * pygfdm is py2-era: it uses `np.complex`, removed in numpy >= 1.24.  We alias
    `np.complex = complex` before importing (an alias, not a re-implementation).
  * pygfdm/filters.py does `import commpy` at module level; scikit-commpy is
    not installed and cannot be.  An EMPTY placeholder module satisfies the
    import statement; no commpy function exists in it and none is called:
    the filter taps are produced by gfdm_amd.filters (taps are an input to the
    kernels, both sides get the same values).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference/python"

np.complex = complex                                  # removed numpy alias used by pygfdm
sys.modules.setdefault("commpy", types.ModuleType("commpy"))   # empty placeholder, see docstring
sys.path.insert(0, REFERENCE)
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))

from pygfdm.gfdm_modulation import gfdm_modulate_block          # noqa: E402
from pygfdm.gfdm_receiver import gfdm_demodulate_block          # noqa: E402
from pygfdm.mapping import get_data_matrix, get_subcarrier_map  # noqa: E402
from gfdm_amd.filters import get_frequency_domain_filter        # noqa: E402

# name, M (timeslots), K (subcarriers), L (overlap), alpha, blocks, active subcarriers (None = all), origin
CASES = [
    ("ref_m16_k4",        16,   4, 2, 0.35, 3, None, "qa_python_bindings.py:254-273,321-341"),
    ("ref_m21_k128",      21, 128, 2, 0.35, 2, None, "qa_python_bindings.py:275-294,343-386"),
    ("ref_m5_k32_a35",     5,  32, 2, 0.35, 4, None, "qa_python_bindings.py:388-440"),
    ("ref_m8_k4",          8,   4, 2, 0.50, 3, None, "qa_simple_modulator_cc.py:39-70, qa_simple_receiver_cc.py:38-57"),
    ("ref_m127_k16_l4",  127,  16, 4, 0.50, 5, None, "qa_simple_modulator_cc.py:72-97"),
    ("ref_m127_k16_l2",  127,  16, 2, 0.50, 5, None, "qa_simple_receiver_cc.py:59-83, qa_advanced_receiver_sb_cc.py:45-82"),
    ("ref_m9_k64_a100",    9,  64, 2, 1.00, 2, None, "qa_advanced_receiver_sb_cc.py:84-119"),
    ("ref_m9_k32_act20",   9,  32, 2, 0.50, 2, 20,   "qa_advanced_receiver_sb_cc.py:134-172"),
    ("cfg1_k32_m5",        5,  32, 2, 0.50, 4, None, "BASELINE.json configs[0]"),
    ("cfg2_k64_m9",        9,  64, 2, 0.20, 4, None, "BASELINE.json configs[1,2]"),
    ("cfg2_k64_m9_act52",  9,  64, 2, 0.20, 2, 52,   "examples/gfdm_simulation_demo.grc:175-180 (Kon=52, dc_free)"),
    ("cfg4_k128_m15_l4",  15, 128, 4, 0.20, 3, None, "BASELINE.json configs[3]"),
    ("cfg4_k128_m15_l4_a50", 15, 128, 4, 0.50, 2, None, "BASELINE.json configs[3], alpha of qa_simple_modulator_cc.py:76"),
    ("cfg5_k256_m31",     31, 256, 2, 0.10, 2, None, "BASELINE.json configs[4]"),
]

CHANNEL = np.array([1.0, 0.5, 0.1j, 0.1 + 0.05j])      # python/qa_python_bindings.py:468


def qpsk(rng, n):
    bits = rng.integers(0, 2, size=(2, n))
    return ((1.0 - 2.0 * bits[0]) + 1j * (1.0 - 2.0 * bits[1])) / np.sqrt(2.0)


def make_case(name, M, K, L, alpha, blocks, active, origin, seed):
    rng = np.random.default_rng(seed)
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    N = M * K
    if active is None:
        smap = np.arange(K)
    else:
        smap = get_subcarrier_map(K, active, dc_free=(active == 52))
    symbols = np.zeros((blocks, K, M), dtype=np.complex128)
    for b in range(blocks):
        symbols[b, smap, :] = qpsk(rng, len(smap) * M).reshape(len(smap), M)
    symbols = symbols.reshape(blocks, N)
    mod = np.empty((blocks, N), dtype=np.complex128)
    for b in range(blocks):
        D = get_data_matrix(symbols[b], K, group_by_subcarrier=False)
        mod[b] = gfdm_modulate_block(D, taps, M, K, L, False)
    out = dict(M=M, K=K, L=L, alpha=alpha, seed=seed, taps=taps, smap=smap.astype(np.int32),
               symbols=symbols, pygfdm_modulate=mod, origin=np.array(origin))
    # a second, non-QPSK input (gaussian) exercises the modulator away from the constellation
    gauss = (rng.standard_normal((blocks, N)) + 1j * rng.standard_normal((blocks, N)))
    gmod = np.empty_like(gauss)
    for b in range(blocks):
        gmod[b] = gfdm_modulate_block(get_data_matrix(gauss[b], K, False), taps, M, K, L, False)
    out["gauss_symbols"] = gauss
    out["pygfdm_modulate_gauss"] = gmod
    if L == 2:   # pygfdm's receiver is only valid for overlap 2 (python/pygfdm/gfdm_receiver.py:54,207)
        dem = np.empty((blocks, N), dtype=np.complex128)
        gdem = np.empty((blocks, N), dtype=np.complex128)
        for b in range(blocks):
            dem[b] = gfdm_demodulate_block(mod[b], taps, K, M, L)
            gdem[b] = gfdm_demodulate_block(gauss[b], taps, K, M, L)   # receiver on arbitrary input
        out["pygfdm_demodulate"] = dem
        out["pygfdm_demodulate_gauss"] = gdem
    # per-block channel for the one-tap equaliser path: FFT_N of the 4-tap test channel times a block phase
    Hc = np.fft.fft(CHANNEL, N)
    f_eq = np.stack([Hc * np.exp(1j * 0.01 * b) for b in range(blocks)])
    out["f_eq"] = f_eq
    out["frame_through_channel"] = np.fft.ifft(np.fft.fft(mod, axis=-1) * f_eq, axis=-1)
    return out


def main():
    for idx, case in enumerate(CASES):
        data = make_case(*case, seed=0x6FD1 + idx)
        path = os.path.join(HERE, case[0] + ".npz")
        np.savez_compressed(path, **data)
        print("%-24s N=%5d blocks=%d  -> %s (%d KiB)" % (case[0], data["M"] * data["K"], case[5], os.path.basename(path),
                                                    os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
