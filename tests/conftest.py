"""pytest configuration: markers, import paths, shared fixtures.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI/pybind surface (no compute on a GPU).
`-m gpu`:       parity of the HIP path (called through the C-ABI and the pybind11 module) with the oracle.
Nothing here or in the gpu tests reads /root/reference.
"""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "gr-gfdm_amd", "python"), os.path.join(ROOT, "gr-gfdm_amd", "lib"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _jit_inside_the_constructor():
    """Run-time instantiated shapes compile inside the constructor during the tests (the library's default moves long compiles to a
    background thread, which would make the kernel family a test runs on depend on timing); the background modes have their own test."""
    try:
        import gfdm_amd
        prev = gfdm_amd.set_jit(gfdm_amd.JIT_IN_CONSTRUCTOR)
    except Exception:
        yield
        return
    yield
    gfdm_amd.set_jit(prev)


def golden_names():
    """kernel-path fixtures (make_golden.py); the composite-transmitter fixtures (make_golden_tx.py) are tx_*"""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if not n.startswith(("tx_", "est_", "ic_", "snr_", "rxl_"))]


def ic_golden_names():
    """interference-cancellation fixtures (make_golden_ic.py): pygfdm's gfdm_get_ic_f_taps / gfdm_remove_sc_interference and
    the composed decide -> cancel -> to_td loop"""
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "ic_*.npz")))


def load_ic_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    for k in ("M", "K", "L", "seed"):
        g[k] = int(g[k])
    return g


def est_golden_names():
    """channel-estimator fixtures (make_golden_est.py)"""
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "est_*.npz")))


def load_est_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    for k in ("M", "K", "A"):
        g[k] = int(g[k])
    return g


def snr_golden_names():
    """estimate_snr known answers (make_golden_snr.py: the reference test's 4 dB case and pygfdm.simulation.estimate_snr0)"""
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "snr_*.npz")))


def load_snr_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    for k in ("M", "K", "A"):
        g[k] = int(g[k])
    g["snr_db"] = float(g["snr_db"])
    return g


def rx_overlap_golden_names():
    """receiver at any overlap against pygfdm's overlap-generic model gfdm_demodulate_fft_loop (make_golden_rx_overlap.py)"""
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "rxl_*.npz")))


def load_rx_overlap_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    for k in ("M", "K", "L"):
        g[k] = int(g[k])
    return g


def tx_golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "tx_*.npz")))


def load_tx_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    for k in ("M", "K", "A", "L", "cp", "cs", "ramp"):
        g[k] = int(g[k])
    g["per_timeslot"] = bool(g["per_timeslot"])
    return g


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    for k in ("M", "K", "L", "seed"):
        g[k] = int(g[k])
    g["alpha"] = float(g["alpha"])
    return g


@pytest.fixture(params=golden_names())
def golden(request):
    return load_golden(request.param)


def rel_err(a, b):
    """worst per-block relative L2 error of a against b (last axis = block)"""
    a = np.asarray(a).reshape(-1, np.asarray(b).shape[-1]) if np.asarray(b).ndim > 1 else np.asarray(a)[None]
    b = np.asarray(b).reshape(a.shape)
    num = np.linalg.norm(a - b, axis=-1)
    den = np.linalg.norm(b, axis=-1)
    return float(np.max(num / np.maximum(den, 1e-30)))


def check_err(tag, err, tol):
    """assert err < tol; with GFDM_ERRLOG=<file> the measured error is also appended there ("tag err tol"), which is where the per-path
    error table profiles/r03/parity_error_table.md comes from (scratch/errlog_table.py)."""
    log = os.environ.get("GFDM_ERRLOG")
    if log:
        with open(log, "a") as f:
            f.write("%s %.3e %.1e\n" % (tag, err, tol))
    assert err < tol, "%s: error %.3e exceeds %.1e" % (tag, err, tol)


def max_abs_component(a, b):
    d = np.asarray(a) - np.asarray(b)
    return float(max(np.max(np.abs(d.real)), np.max(np.abs(d.imag))))


def assert_places(a, b, places):
    """gr_unittest.assertComplexTuplesAlmostEqual semantics: every |d re|, |d im| rounds to 0 at `places` decimals."""
    assert max_abs_component(a, b) < 0.5 * 10.0 ** (-places), "max component error %.3e" % max_abs_component(a, b)


def have_gpu():
    try:
        import gfdm_amd
        return gfdm_amd.lib().gfdm_hip_device_count() > 0
    except Exception:
        return False
