#!/usr/bin/env python3
"""Golden vectors for the interference-cancellation (IC) stage, from the reference's own Python model.

Runs ONLY in the build container (needs /root/reference); the *.npz it writes are data (seeded inputs + the reference
model's outputs).  No reference source or bytecode is written anywhere (sys.dont_write_bytecode).

Which reference functions pin what (python/pygfdm/gfdm_receiver.py):
  * gfdm_get_ic_f_taps(f_taps, M)                  :99-100   == receiver_kernel_cc::ic_filter_taps
                                                               (lib/receiver_kernel_cc.cc:56-63)   ic[m] = t[m] t[(L-1)M + m]
  * gfdm_remove_sc_interference(R, D, K, M, L, H)  :108-114  == receiver_kernel_cc::cancel_sc_interference
                                                               (lib/receiver_kernel_cc.cc:274-299) once H_sic is passed in
  * gfdm_transform_subcarriers_to_tdomain          :91-96    == transform_subcarriers_to_td (:211-225)
  * gfdm_map_subcarriers -> utils.map_qpsk_stream  :103-105, utils.py:80-82 == the QPSK decision of
                                                               advanced_receiver_kernel_cc::map_symbols_to_constellation_points
                                                               (lib/advanced_receiver_kernel_cc.cc:109-123) away from ties
                                                               (np.sign(0) = 0 there, the constellation maps 0 to the negative point)
Only gfdm_demodulate_block_sic (:117-138) is NOT usable as a golden: it scales the taps by 1/K (:127) and prints; the loop
below is the same composition (:131-137) with the C++ scaling, i.e. H_sic from the energy-M-normalised taps, and with S
not updated between rounds, exactly lib/advanced_receiver_kernel_cc.cc:56-76.

Matrices of these functions are [M][K] (column k = subcarrier k), the transpose of the kernels' [k][m] block layout.

Import notes: `np.complex = complex` and `builtins.xrange = range` are aliases for names that pygfdm (py2-era) uses and
modern Python / numpy dropped (gfdm_receiver.py:39,112); `commpy` is satisfied by an EMPTY placeholder module (imported
at module level by pygfdm/filters.py, none of its functions is called; taps come from gfdm_amd.filters and are an input
to both sides).
"""
import builtins
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference/python"

np.complex = complex
builtins.xrange = range
sys.modules.setdefault("commpy", types.ModuleType("commpy"))
sys.path.insert(0, REFERENCE)
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from pygfdm import gfdm_receiver as PR                           # noqa: E402
from pygfdm.gfdm_modulation import gfdm_modulate_block           # noqa: E402
from pygfdm.mapping import get_data_matrix                       # noqa: E402
from pygfdm.utils import map_qpsk_stream                         # noqa: E402
from gfdm_amd.filters import get_frequency_domain_filter         # noqa: E402

# name, M, K, L, alpha, blocks, origin
CASES = [
    ("ic_ref_m5_k32_a35",    5,  32, 2, 0.35, 4, "qa_python_bindings.py:388-415 (genie IC shape)"),
    ("ic_ref_m9_k64_a100",   9,  64, 2, 1.00, 2, "qa_advanced_receiver_sb_cc.py:84-119"),
    ("ic_ref_m9_k32_a50",    9,  32, 2, 0.50, 2, "qa_advanced_receiver_sb_cc.py:134-172 (shape; all subcarriers active: the Python model has no subcarrier map)"),
    ("ic_cfg1_k32_m5",       5,  32, 2, 0.50, 4, "BASELINE.json configs[0]"),
    ("ic_cfg2_k64_m9",       9,  64, 2, 0.20, 4, "BASELINE.json configs[1,2]"),
    ("ic_cfg4_k128_m15_l4", 15, 128, 4, 0.20, 2, "BASELINE.json configs[3]"),
    ("ic_cfg5_k256_m31",    31, 256, 2, 0.10, 1, "BASELINE.json configs[4]"),
    ("ic_m21_k12",          21,  12, 2, 0.35, 3, "generic-family shape of smoke()"),
]
ROUNDS = 5
TIE_GUARD = 1e-3      # a case whose decided values come closer than this to a decision boundary is rejected (re-seed)


def normalise(taps, M):
    """the constructor's tap normalisation (lib/receiver_kernel_cc.cc:99-118); an INPUT of the functions under test"""
    t = np.asarray(taps, dtype=np.complex128)
    return t / np.sqrt(abs(np.sum(t * np.conj(t))) / M)


def to_mat(v, K, M):
    """[k][m] block vector -> the [M][K] matrix pygfdm works on"""
    return np.reshape(v, (K, M)).T


def from_mat(A):
    return A.T.reshape(-1)


def qpsk(rng, n):
    bits = rng.integers(0, 2, size=(2, n))
    return ((1.0 - 2.0 * bits[0]) + 1j * (1.0 - 2.0 * bits[1])) / np.sqrt(2.0)


def pygfdm_fd(frame, taps, K, M, L):
    """S of gfdm_demodulate_block (:117-124), overlap 2 only"""
    D0 = PR.gfdm_transform_input_to_fd(frame)
    D1 = PR.gfdm_extract_subcarriers(D0, K, M, L)
    D2 = PR.gfdm_filter_subcarriers(D1, taps, K, M, L)
    return PR.gfdm_superposition_subcarriers(D2, K, M, L)       # [M][K]


def make_case(name, M, K, L, alpha, blocks, origin, seed):
    rng = np.random.default_rng(seed)
    N = M * K
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    nt = normalise(taps, M)
    ic_taps = PR.gfdm_get_ic_f_taps(nt, M)
    # a second tap set, complex and asymmetric: exercises the general (non real-symmetric) cancellation kernel
    ctaps = nt * np.exp(1j * rng.uniform(-np.pi, np.pi, L * M)) * rng.uniform(0.5, 1.5, L * M)
    cnt = normalise(ctaps, M)
    ic_ctaps = PR.gfdm_get_ic_f_taps(cnt, M)

    out = dict(M=M, K=K, L=L, alpha=alpha, seed=seed, taps=taps, ctaps=ctaps, origin=np.array(origin),
               pygfdm_ic_taps=ic_taps, pygfdm_ic_ctaps=ic_ctaps)

    # (1) cancel_sc_interference on arbitrary (gaussian) inputs, both tap sets
    td = rng.standard_normal((blocks, N)) + 1j * rng.standard_normal((blocks, N))
    fd = rng.standard_normal((blocks, N)) + 1j * rng.standard_normal((blocks, N))
    out["td_in"], out["fd_in"] = td, fd
    out["pygfdm_cancel"] = np.stack([from_mat(PR.gfdm_remove_sc_interference(to_mat(fd[b], K, M), to_mat(td[b], K, M), K, M, L, ic_taps))
                                     for b in range(blocks)])
    out["pygfdm_cancel_ctaps"] = np.stack([from_mat(PR.gfdm_remove_sc_interference(to_mat(fd[b], K, M), to_mat(td[b], K, M), K, M, L, ic_ctaps))
                                           for b in range(blocks)])

    # (2) the IC loop on a modulated frame: S -> d0 -> ROUNDS x (decide, cancel, to_td)
    symbols = qpsk(rng, blocks * N).reshape(blocks, N)
    frames = np.stack([gfdm_modulate_block(get_data_matrix(symbols[b], K, False), taps, M, K, L, False) for b in range(blocks)])
    out["symbols"], out["frames"] = symbols, frames
    if L == 2:
        S = np.stack([from_mat(pygfdm_fd(frames[b], nt, K, M, L)) for b in range(blocks)])
        out["S_source"] = np.array("pygfdm (gfdm_receiver.py:117-124)")
    else:
        # pygfdm's subcarrier extraction hard-codes overlap 2 (:54); S is an INPUT of the IC stage here, taken from the numpy
        # oracle (whose overlap != 2 receiver is pinned by the transpose identity, tests/test_oracle.py)
        import gfdm_ref
        S = gfdm_ref.fft_filter_downsample(frames, nt, M, K, L)
        out["S_source"] = np.array("oracle/gfdm_ref.py fft_filter_downsample (input of the IC stage)")
    out["S"] = S
    d = np.stack([from_mat(PR.gfdm_transform_subcarriers_to_tdomain(to_mat(S[b], K, M), K, M, L)) for b in range(blocks)])
    out["pygfdm_d0"] = d
    iters, margins = [], []
    for _ in range(ROUNDS):
        margins.append(min(np.min(np.abs(d.real)), np.min(np.abs(d.imag))))
        nxt = np.empty_like(d)
        for b in range(blocks):
            dec = np.reshape(np.asarray(map_qpsk_stream(to_mat(d[b], K, M))), (M, K))           # gfdm_map_subcarriers (:103-105)
            fdn = PR.gfdm_remove_sc_interference(to_mat(S[b], K, M), dec, K, M, L, ic_taps)     # S is NOT updated between rounds
            nxt[b] = from_mat(PR.gfdm_transform_subcarriers_to_tdomain(fdn, K, M, L))
        d = nxt
        iters.append(d.copy())
    out["pygfdm_ic_iters"] = np.stack(iters)                    # [round][block][N]
    out["decision_margin"] = np.array(margins)
    return out


def main():
    for idx, case in enumerate(CASES):
        seed = 0x1C00 + idx
        while True:
            data = make_case(*case, seed=seed)
            if data["decision_margin"].min() > TIE_GUARD:
                break
            seed += 1000                                         # a decided value sits on a boundary: take another seed
        path = os.path.join(HERE, case[0] + ".npz")
        np.savez_compressed(path, **data)
        print("%-22s N=%5d blocks=%d margin=%.4f seed=%#x -> %s (%d KiB)" % (
            case[0], data["M"] * data["K"], case[5], data["decision_margin"].min(), seed, os.path.basename(path),
            os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
