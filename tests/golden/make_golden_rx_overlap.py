#!/usr/bin/env python3
"""Golden vectors that pin the RECEIVER's filter stage for overlap != 2 (lib/receiver_kernel_cc.cc:165-192, 301-334) to the reference.

pygfdm's `gfdm_demodulate_block` -- the model make_golden.py uses -- is only valid for overlap 2 (python/pygfdm/gfdm_receiver.py:54,207;
its own docstring says so).  But the same file holds a second, overlap-generic receiver model that the reference compares it with:

    gfdm_demodulate_fft_loop(rx, timeslots, subcarriers, overlap, sparse_freq_taps)        python/pygfdm/gfdm_receiver.py:190-199

(per subcarrier: roll the spectrum to the subcarrier's centre, keep overlap * timeslots bins, multiply by the sparse taps / subcarriers,
fold the overlap segments, inverse FFT).  Its `/ subcarriers` is a scaling convention of that function; with it undone the model equals
`gfdm_demodulate_block` exactly for overlap 2 (checked here) and gives the reference's receiver for every other overlap.  Stored:
`pygfdm_demodulate_fft_loop` = subcarriers * model output, for modulated QPSK frames and for arbitrary (gaussian) input.

Build container only (imports /root/reference/python/pygfdm); import notes as in make_golden.py.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
np.complex = complex
sys.modules.setdefault("commpy", types.ModuleType("commpy"))
sys.path.insert(0, "/root/reference/python")
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))

from pygfdm.gfdm_modulation import gfdm_modulate_block                                   # noqa: E402
from pygfdm.gfdm_receiver import gfdm_demodulate_block, gfdm_demodulate_fft_loop         # noqa: E402
from pygfdm.mapping import get_data_matrix                                               # noqa: E402
from gfdm_amd.filters import get_frequency_domain_filter                                 # noqa: E402

# name, M, K, L, alpha, blocks, origin
CASES = [
    ("rxl_cfg4_k128_m15_l4", 15, 128, 4, 0.2, 2, "BASELINE.json configs[3]"),
    ("rxl_cfg4_k128_m15_l4_a50", 15, 128, 4, 0.5, 2, "BASELINE.json configs[3], alpha of qa_simple_modulator_cc.py:76"),
    ("rxl_ref_m127_k16_l4", 127, 16, 4, 0.5, 2, "qa_simple_modulator_cc.py:72-97 (the reference's only overlap-4 shape)"),
    ("rxl_m7_k16_l6", 7, 16, 6, 0.3, 3, "overlap 6"),
    ("rxl_m9_k32_l8", 9, 32, 8, 0.4, 3, "overlap 8"),
    ("rxl_m21_k12_l4", 21, 12, 4, 0.35, 2, "overlap 4, subcarriers not a power of two"),
    ("rxl_cfg2_k64_m9_l2", 9, 64, 2, 0.2, 2, "overlap 2: the two models of the reference against each other"),
]


def main():
    for idx, (name, M, K, L, alpha, blocks, origin) in enumerate(CASES):
        rng = np.random.default_rng(0xF1F7 + idx)
        taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
        N = M * K
        bits = rng.integers(0, 2, size=(2, blocks, N))
        symbols = ((1.0 - 2.0 * bits[0]) + 1j * (1.0 - 2.0 * bits[1])) / np.sqrt(2.0)
        frames = np.array([gfdm_modulate_block(get_data_matrix(symbols[b], K, group_by_subcarrier=False), taps, M, K, L, False) for b in range(blocks)])
        gauss = rng.standard_normal((blocks, N)) + 1j * rng.standard_normal((blocks, N))
        # the inputs are stored in single precision (what the kernels take); the model runs on exactly those values
        frames, gauss = frames.astype(np.complex64).astype(np.complex128), gauss.astype(np.complex64).astype(np.complex128)
        dem = np.array([K * gfdm_demodulate_fft_loop(frames[b], M, K, L, taps) for b in range(blocks)])
        gdem = np.array([K * gfdm_demodulate_fft_loop(gauss[b], M, K, L, taps) for b in range(blocks)])
        if L == 2:                        # the model make_golden.py uses says the same
            other = np.array([gfdm_demodulate_block(gauss[b], taps, K, M, L) for b in range(blocks)])
            assert np.max(np.abs(other - gdem)) < 1e-12 * np.max(np.abs(gdem))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), M=M, K=K, L=L, alpha=alpha, taps=taps, frames=frames.astype(np.complex64), gauss=gauss.astype(np.complex64),
                            pygfdm_demodulate_fft_loop=dem, pygfdm_demodulate_fft_loop_gauss=gdem, origin=np.array(origin))
        print("%-28s N=%5d blocks=%d" % (name, N, blocks))


if __name__ == "__main__":
    main()
