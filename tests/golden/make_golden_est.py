#!/usr/bin/env python3
"""Golden vectors for the preamble channel estimator (preamble_channel_estimator_cc), produced
with the reference's own Python model of it:

    pygfdm.validation_utils.frame_estimator.estimate_frame   (python/pygfdm/validation_utils.py:33-81)

and preambles composed like python/qa_channel_estimator_cc.py:62-110 does (mapped_preamble = random QPSK on the active
subcarriers -> get_sync_symbol; the filter taps come from gfdm_amd.filters because pygfdm's need commpy, as in
make_golden.py).  The Python model covers the dc-free configuration, the one every reference test and flowgraph uses
(qa_channel_estimator_cc.py:73-74,103-104, examples/hier_gfdm_receiver.grc).  pygfdm/simulation.py's SNR estimators are not
importable here (they import the installed `gfdm` package), so estimate_snr has no golden vector: its tests are properties.

Build container only (imports /root/reference/python/pygfdm); see make_golden.py for the import notes.  One more alias is
needed here: scipy removed `scipy.signal.gaussian`; it is aliased to `scipy.signal.windows.gaussian` (same function).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import scipy.signal as signal

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
np.complex = complex
sys.modules.setdefault("commpy", types.ModuleType("commpy"))
if not hasattr(signal, "gaussian"):
    signal.gaussian = signal.windows.gaussian
sys.path.insert(0, "/root/reference/python")
sys.path.insert(0, os.path.join(ROOT, "gr-gfdm_amd", "python"))

from pygfdm.mapping import get_subcarrier_map, map_to_waveform_resources       # noqa: E402
from pygfdm.preamble import get_sync_symbol                                    # noqa: E402
from pygfdm.utils import calculate_signal_energy, get_random_qpsk              # noqa: E402
from pygfdm.validation_utils import frame_estimator                            # noqa: E402
from gfdm_amd.filters import get_frequency_domain_filter                       # noqa: E402

SEED = int(3660365253)                                                         # qa_channel_estimator_cc.py:56
# name, M, K, active, alpha, origin
CASES = [
    ("est_ref_m3_k32_a24", 3, 32, 24, 0.5, "qa_channel_estimator_cc.py:61-85 (test_001_simple)"),
    ("est_ref_m5_k64_a52", 5, 64, 52, 0.5, "qa_channel_estimator_cc.py:87-125 (test_002_selective)"),
    ("est_cfg2_m9_k64_a52", 9, 64, 52, 0.2, "BASELINE.json configs[1] with examples/gfdm_simulation_demo.grc Kon=52"),
    ("est_cfg4_m15_k128_a110", 15, 128, 110, 0.2, "BASELINE.json configs[3]"),
]


def main():
    for name, M, K, A, alpha, origin in CASES:
        rng = np.random.default_rng(sum(map(ord, name)))
        L, cp, ramp = 2, K // 2, K // 4
        smap = get_subcarrier_map(K, A, dc_free=True)
        pn_sym = map_to_waveform_resources(get_random_qpsk(A, SEED), A, K, smap)
        H = get_frequency_domain_filter("rrc", alpha, 2, K, L)
        H = H / np.sqrt(calculate_signal_energy(H) / 2.0)                      # generate_sync_symbol, preamble.py:128-132
        full, core = get_sync_symbol(pn_sym, H, K, L, cp, ramp)
        h = np.array([1., .5, .1j, .1 + .05j])
        through = np.convolve(full, h, "full")[:full.size][cp:-ramp]           # qa_channel_estimator_cc.py:104-106
        rx = [core, through]
        for snr_db in (30.0, 10.0):
            sigma = np.sqrt(np.mean(np.abs(through) ** 2) / 10 ** (snr_db / 10) / 2)
            rx.append(through + sigma * (rng.standard_normal(2 * K) + 1j * rng.standard_normal(2 * K)))
        rx = np.array(rx)
        est = frame_estimator(core, K, M, A)
        frames = np.array([est.estimate_frame(r) for r in rx])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), M=M, K=K, A=A, preamble=core, rx_preambles=rx, channel=h,
                            smap=smap, pygfdm_frame_estimates=frames, noise_snr_db=np.array([30.0, 10.0]), origin=origin)
        print(name, rx.shape, frames.shape)


if __name__ == "__main__":
    main()
