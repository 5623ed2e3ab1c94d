#!/usr/bin/env python3
"""Golden vectors for the composite transmitter (gr-gfdm transmitter_kernel: resource mapper -> modulator ->
cyclic prefix/suffix + window ramp -> preamble), produced with the reference's Python model exactly the way the
reference's own test composes its expectation (python/qa_transmitter_cc.py:42-55):

    dd    = map_to_waveform_resources(symbols, active, K, smap[, per_timeslot])
    b     = gfdm_modulate_block(get_data_matrix(dd, K, False), taps, M, K, L, False)
    frame = concatenate(preamble, pinch_block(add_cyclic_starfix(roll(b, shift), cp, cs), window))   per cyclic shift

Build container only (imports /root/reference/python/pygfdm); see make_golden.py for the import notes.  The preambles
are arbitrary complex sequences (they are only copied in front of the frame); the reference's preamble generator needs
the absent commpy package.
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

from pygfdm.cyclic_prefix import add_cyclic_starfix, get_raised_cosine_ramp, get_window_len, pinch_block   # noqa: E402
from pygfdm.gfdm_modulation import gfdm_modulate_block                                                      # noqa: E402
from pygfdm.mapping import get_data_matrix, get_subcarrier_map, map_to_waveform_resources                   # noqa: E402
from gfdm_amd.filters import get_frequency_domain_filter                                                    # noqa: E402

# name, M, K, active, dc_free, L, alpha, cp, cs, ramp, per_timeslot, cyclic shifts, preamble length, frames, symbols per frame (None = full)
CASES = [
    ("tx_ref_k64_m9_cdd",   9,  64,  52, True,  2, 0.5, 16,  8,  8, True,  [0, 3, 7, 8], 160, 3, None),   # qa_transmitter_cc.py:80-183
    ("tx_k64_m9_persc",     9,  64,  52, True,  2, 0.2, 16,  8,  4, False, [0, 5],       128, 2, None),
    ("tx_k32_m5_short",     5,  32,  20, False, 2, 0.5,  8,  4,  4, True,  [0],           64, 3, 77),     # fewer symbols than slots: zero padded
    ("tx_k128_m15_l4",     15, 128, 100, True,  4, 0.2, 32, 16, 16, True,  [0, 9],       256, 2, None),
]


def main():
    for idx, (name, M, K, A, dc_free, L, alpha, cp, cs, ramp, per_ts, shifts, plen, frames, nsym) in enumerate(CASES):
        rng = np.random.default_rng(0x7F00 + idx)
        taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
        smap = get_subcarrier_map(K, A, dc_free=dc_free)
        window = get_raised_cosine_ramp(ramp, get_window_len(cp, M, K, cs))
        preambles = [(rng.standard_normal(plen) + 1j * rng.standard_normal(plen)) for _ in shifts]
        n_in = A * M if nsym is None else nsym
        symbols = ((1 - 2 * rng.integers(0, 2, (frames, n_in))) + 1j * (1 - 2 * rng.integers(0, 2, (frames, n_in)))) / np.sqrt(2)
        out = [[] for _ in shifts]
        blocks = []
        for f in range(frames):
            dd = map_to_waveform_resources(symbols[f], A, K, smap, per_ts)
            b = gfdm_modulate_block(get_data_matrix(dd, K, False), taps, M, K, L, False)
            blocks.append(b)
            for i, (s, p) in enumerate(zip(shifts, preambles)):
                data = pinch_block(add_cyclic_starfix(np.roll(b, s), cp, cs), window)
                out[i].append(np.concatenate((p, data)))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), M=M, K=K, A=A, L=L, alpha=alpha, cp=cp, cs=cs, ramp=ramp,
                            per_timeslot=per_ts, shifts=np.array(shifts, np.int32), smap=np.array(smap, np.int32), taps=taps,
                            window=window.astype(np.complex128), preambles=np.array(preambles), symbols=symbols,
                            pygfdm_blocks=np.array(blocks), pygfdm_frames=np.array(out))
        print(name, "frame", out[0][0].size, "ports", len(shifts))


if __name__ == "__main__":
    main()


def add_rx_expectations():
    """Receiver-side chain on the transmitter frames, with the reference model where it is
    valid (overlap 2): frame (port 0, preamble stripped) -> remove cyclic prefix (slice, cyclic_prefix.py) ->
    gfdm_demodulate_block -> demap_from_waveform_resource_grid (per-timeslot order, mapping.py:56-59).  Stored next to the
    transmitter vectors as `pygfdm_rx_symbols`."""
    from pygfdm.gfdm_receiver import gfdm_demodulate_block
    from pygfdm.mapping import demap_from_waveform_resource_grid
    for name, M, K, A, dc_free, L, alpha, cp, cs, ramp, per_ts, shifts, plen, frames, nsym in CASES:
        path = os.path.join(HERE, name + ".npz")
        z = dict(np.load(path))
        if L != 2 or not per_ts:
            continue
        rx = []
        for f in range(frames):
            body = z["pygfdm_frames"][0][f][plen:]                      # window ramp only touches prefix / suffix samples here
            block = body[cp:cp + M * K]
            d = gfdm_demodulate_block(block, z["taps"], K, M, L)
            rx.append(demap_from_waveform_resource_grid(d, K, z["smap"]))
        z["pygfdm_rx_symbols"] = np.array(rx)
        np.savez_compressed(path, **z)
        print(name, "rx symbols", z["pygfdm_rx_symbols"].shape)


if __name__ == "__main__":
    add_rx_expectations()


def add_stage_expectations():
    """The stand-alone stages either side of the modulator / receiver (Resource_mapper, Cyclic_prefixer of the reference's Python
    bindings), with the reference model:
      pygfdm_grid      map_to_waveform_resources(symbols, A, K, smap, per_timeslot)       mapping.py:53-55,64-76   [frames][K*M]
      grid_in          random [frames][K*M] grids (own seed)
      pygfdm_demapped  demap_from_waveform_resource_grid(grid_in, K, smap)               mapping.py:58-61 (per-timeslot order)
    The cyclic-prefix stage needs no new arrays: pygfdm_blocks -> pygfdm_frames[port][frame][preamble length:] is
    pinch_block(add_cyclic_starfix(roll(block, shift), cp, cs), window), stored since round 1."""
    from pygfdm.mapping import demap_from_waveform_resource_grid
    for idx, (name, M, K, A, dc_free, L, alpha, cp, cs, ramp, per_ts, shifts, plen, frames, nsym) in enumerate(CASES):
        path = os.path.join(HERE, name + ".npz")
        z = dict(np.load(path))
        rng = np.random.default_rng(0x7F80 + idx)
        z["pygfdm_grid"] = np.array([map_to_waveform_resources(z["symbols"][f], A, K, z["smap"], per_ts) for f in range(frames)])
        grid_in = rng.standard_normal((frames, K * M)) + 1j * rng.standard_normal((frames, K * M))
        z["grid_in"] = grid_in
        z["pygfdm_demapped"] = np.array([demap_from_waveform_resource_grid(grid_in[f], K, z["smap"]) for f in range(frames)])
        np.savez_compressed(path, **z)
        print(name, "grid", z["pygfdm_grid"].shape, "demapped", z["pygfdm_demapped"].shape)


if __name__ == "__main__":
    add_stage_expectations()
