"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/gfdm_hip.h
declares, argument validation mirrors the reference constructors, the pybind11 module has the reference's surface,
and there is no CPU fallback (without a GPU every handle creation fails loudly)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, have_gpu

HEADER = os.path.join(ROOT, "include", "gfdm_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gfdm_hip_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import gfdm_amd
    lib = ctypes.CDLL(gfdm_amd.capi.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 38
    for n in names:
        assert hasattr(lib, n), "libgfdm_hip.so does not export %s" % n
    assert set(gfdm_amd.exported_symbols()) == set(names), "ctypes binding and header disagree"


def test_header_is_plain_c():
    """the boundary is a C ABI: include/gfdm_hip.h must compile as C99 on its own (plain pointers and sizes, no C++ or torch types)"""
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", HEADER])


@pytest.mark.gpu
def test_c_program_runs_a_block_on_the_gpu(tmp_path):
    """the same C program where a GPU is present: error codes and a block through the modulator and the receiver"""
    if not have_gpu():
        pytest.fail("no MI355X visible")
    test_c_program_links_and_calls_the_boundary(tmp_path)


def test_c_program_links_and_calls_the_boundary(tmp_path):
    """tests/c_abi_smoke.c, compiled as C99 against include/gfdm_hip.h and linked with libgfdm_hip.so only: runs here (no GPU: every
    create must answer GFDM_HIP_ENODEV) and on the GPU box (a block through the modulator and the receiver)."""
    import subprocess
    import gfdm_amd
    libdir = os.path.dirname(gfdm_amd.capi.LIB_PATH)
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi_smoke.c"),
                           "-L" + libdir, "-lgfdm_hip", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert ("no device" in out.stdout) != have_gpu()


def test_strerror_and_version():
    import gfdm_amd
    L = gfdm_amd.lib()
    assert L.gfdm_hip_strerror(0) == b"success"
    assert b"overlap" in L.gfdm_hip_strerror(-2)
    assert b"gfx950" in L.gfdm_hip_version()


def test_constructor_validation_matches_reference():
    """lib/modulator_kernel_cc.cc:39-46, lib/receiver_kernel_cc.cc:40-52: std::invalid_argument -> ValueError."""
    import gfdm_amd
    import gfdm_python
    taps = np.ones(17, np.complex64)
    for cls in (gfdm_amd.Modulator, gfdm_amd.Demodulator, gfdm_python.Modulator, gfdm_python.Demodulator):
        with pytest.raises(ValueError, match=r"number of frequency taps\(17\) MUST be equal to n_timeslots\(9\) \* overlap\(2\) = 18!"):
            cls(9, 64, 2, taps)
    for cls in (gfdm_amd.Demodulator, gfdm_python.Demodulator):
        with pytest.raises(ValueError, match="overlap MUST be greater or equal 2"):
            cls(9, 64, 1, np.ones(9, np.complex64))
    with pytest.raises(ValueError):
        gfdm_python.AdvancedReceiver(9, 64, 2, taps, list(range(64)), 2, gfdm_python.Constellation.qpsk(), 0)


def test_pybind_surface_matches_reference():
    """python/bindings/modulator_python.cc:34-59, demodulator_python.cc:35-205."""
    import gfdm_python
    for name in ("block_size", "filter_taps", "modulate"):
        assert hasattr(gfdm_python.Modulator, name)
    for name in ("timeslots", "subcarriers", "overlap", "block_size", "filter_taps", "demodulate", "fft_filter_downsample",
                 "transform_subcarriers_to_td", "demodulate_equalize", "fft_equalize_filter_downsample", "cancel_sc_interference"):
        assert hasattr(gfdm_python.Demodulator, name)
    for name in ("block_size", "set_ic", "get_ic", "set_phase_compensation", "get_phase_compensation", "demodulate", "demodulate_equalize"):
        assert hasattr(gfdm_python.AdvancedReceiver, name)
    for name in ("block_size", "frame_size", "map_to_resources", "demap_from_resources"):
        assert hasattr(gfdm_python.Resource_mapper, name)                  # resource_mapper_python.cc:30-87
    for name in ("block_size", "frame_size", "cyclic_shift", "add_cyclic_prefix", "remove_cyclic_prefix"):
        assert hasattr(gfdm_python.Cyclic_prefixer, name)                  # cyclic_prefix_python.cc:31-93
    for name in ("timeslots", "subcarriers", "active_subcarriers", "frame_len", "is_dc_free", "estimate_frame", "estimate_snr"):
        assert hasattr(gfdm_python.Preamble_channel_estimator, name)      # preamble_channel_estimator_python.cc:33-99
    q = gfdm_python.Constellation.qpsk()
    pts = np.array(q.points())
    assert np.allclose(pts, np.array([-1 - 1j, 1 - 1j, -1 + 1j, 1 + 1j]) / np.sqrt(2))
    assert [q.decision_maker(p) for p in pts] == [0, 1, 2, 3]
    assert q.decision_maker(0j) == 0                        # zero maps to the negative point (sign test is '> 0')


def test_no_cpu_fallback():
    """Without a GPU the product path must refuse to run (never silently compute on the host)."""
    if have_gpu():
        pytest.skip("a GPU is present")
    import gfdm_amd
    import gfdm_python
    taps = np.ones(18, np.complex64)
    with pytest.raises(gfdm_amd.GfdmHipError) as e:
        gfdm_amd.Modulator(9, 64, 2, taps)
    assert e.value.status == gfdm_amd.capi.ENODEV
    with pytest.raises(RuntimeError, match="no HIP device"):
        gfdm_python.Demodulator(9, 64, 2, taps)


def test_product_sources_do_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under gr-gfdm_amd/ or include/ may reference it."""
    bad = []
    for base in (os.path.join(ROOT, "gr-gfdm_amd"), os.path.join(ROOT, "include")):
        for dirpath, _, files in os.walk(base):
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"gfdm_oracle|gfdm_ref|c_oracle|oracle/", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_reference_pybind_bindings_compile_unchanged():
    """Drop-in evidence: gr-gfdm's own kernel bindings (python/bindings/{modulator,demodulator,preamble_channel_estimator,resource_mapper,
    cyclic_prefix}_python.cc -- all five kernel-level classes of its Python module) compile,
    unmodified and from where they lie, against this repository's class headers.  Build-container only: the
    reference checkout does not exist on the GPU box."""
    import subprocess
    import sysconfig
    import pybind11
    ref = "/root/reference/python/bindings"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present")
    inc = ["-I" + os.path.join(ROOT, "gr-gfdm_amd", "cpp", "include"), "-I" + os.path.join(ROOT, "include"),
           "-I" + sysconfig.get_paths()["include"], "-I" + pybind11.get_include()]
    for name in ("modulator_python.cc", "demodulator_python.cc", "preamble_channel_estimator_python.cc", "resource_mapper_python.cc",
                 "cyclic_prefix_python.cc"):
        subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only"] + inc + [os.path.join(ref, name)])


def test_reference_gnuradio_wrappers_compile_unchanged_on_these_kernel_classes():
    """north_star: "so the GNU Radio *_impl wrappers still drop in" (SURVEY.md section 8(f)4).  gr-gfdm's own block wrappers --
    lib/{simple_modulator,simple_receiver,advanced_receiver_sb,transmitter,channel_estimator,resource_mapper,resource_demapper,
    cyclic_prefixer}_cc_impl.cc, unmodified and from where they lie -- are syntax-checked with THIS repository's kernel class
    headers in front of the reference's include directory (so <gfdm/modulator_kernel_cc.h> etc. are ours, and only the block-interface
    headers such as <gfdm/simple_modulator_cc.h> come from the reference).  GNU Radio itself is absent: tests/mock_gnuradio declares
    (declarations only, nothing links) the block-API names the wrappers use.  Any error naming a kernel-class member would be a
    boundary bug.  Build-container only: the reference checkout does not exist on the GPU box."""
    import subprocess
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "lib")):
        pytest.skip("reference checkout not present")
    ours = os.path.join(ROOT, "gr-gfdm_amd", "cpp", "include")
    inc = ["-I" + ours, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "mock_gnuradio"),
           "-I" + os.path.join(ref, "include"), "-I" + os.path.join(ref, "lib")]
    kernel_headers = {"simple_modulator": "modulator_kernel_cc.h", "simple_receiver": "receiver_kernel_cc.h",
                      "advanced_receiver_sb": "advanced_receiver_kernel_cc.h", "transmitter": "transmitter_kernel.h",
                      "channel_estimator": "preamble_channel_estimator_cc.h", "resource_mapper": "resource_mapper_kernel_cc.h",
                      "resource_demapper": "resource_mapper_kernel_cc.h", "cyclic_prefixer": "add_cyclic_prefix_cc.h"}      # (remove_prefix_cc_impl.cc uses no kernel class)
    for name, hdr in kernel_headers.items():
        src = os.path.join(ref, "lib", name + "_cc_impl.cc")
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-H"] + inc + [src], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        used = [ln.split()[-1] for ln in r.stderr.splitlines() if ln.startswith(".") and ln.endswith("/gfdm/" + hdr)]
        assert used and all(u.startswith(ours) for u in used), (name, used)      # the kernel class really is this repository's


def test_row_lane_kernels_build_through_hiprtc(tmp_path, monkeypatch):
    """Run-time instantiation (gfdm_jit.hip): the kernel headers embedded in libgfdm_hip.so compile through hiprtc for shapes outside
    the compiled list (no GPU needed for the compile), the code object lands in the disk cache and is found there the second time;
    shapes the row-lane layout cannot hold are refused."""
    import time
    import gfdm_amd
    monkeypatch.setenv("GFDM_HIP_CACHE_DIR", str(tmp_path))
    L = gfdm_amd.lib()
    t0 = time.perf_counter()
    for part in range(5):
        assert L.gfdm_hip_jit_build_for_testing(7, 16, 2, part) == 0, L.gfdm_hip_last_error()
    first = time.perf_counter() - t0
    files = sorted(p.name for p in tmp_path.iterdir())
    assert len([f for f in files if f.endswith(".hsaco")]) == 5 and len([f for f in files if f.endswith(".names")]) == 5
    t0 = time.perf_counter()
    for part in range(5):
        assert L.gfdm_hip_jit_build_for_testing(7, 16, 2, part) == 0
    assert time.perf_counter() - t0 < 0.5 * first + 0.2            # served from the cache
    assert L.gfdm_hip_jit_build_for_testing(13, 32, 4, 1) == 0, L.gfdm_hip_last_error()      # overlap 4, IC kernels
    assert L.gfdm_hip_jit_build_for_testing(5, 12, 2, 0) == 0, L.gfdm_hip_last_error()       # K = 12: not a power of two, one radix-12 pass
    assert L.gfdm_hip_jit_build_for_testing(3, 48, 4, 1) == 0, L.gfdm_hip_last_error()       # K = 48 = 3 x 16
    # K with a prime factor above 32 (74 = 2 x 37, the prime 1021) or above 1024, M / K / L out of range
    assert L.gfdm_hip_jit_build_for_testing(7, 16, 2, 5) != 0                                # there is no part 5
    assert L.gfdm_hip_jit_build_for_testing(3, 200, 2, 0) == 0, L.gfdm_hip_last_error()      # K = 200 = 2 x 10 x 10: three passes
    for (M, K, Lp) in ((9, 74, 2), (9, 1021, 2), (9, 1040, 2), (127, 16, 2), (9, 2048, 2), (19, 1024, 2), (2, 16, 2), (9, 64, 1), (49, 64, 2), (9, 6, 8)):
        assert L.gfdm_hip_jit_build_for_testing(M, K, Lp, 0) != 0


def test_precompile_and_library_switches_without_a_gpu(tmp_path, monkeypatch):
    """gfdm_hip_precompile (the deployment step of INTEGRATION.md) fills the code-object cache without a handle or a GPU; the three
    library switches keep their mode and clamp it; python -m gfdm_amd.precompile is the command-line form."""
    import subprocess
    import sys
    import gfdm_amd
    monkeypatch.setenv("GFDM_HIP_CACHE_DIR", str(tmp_path))
    gfdm_amd.precompile(6, 16, 2, 1 | 8)                               # receive + modulate parts of a run-time shape
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]) == 2
    gfdm_amd.precompile(9, 64, 2)                                      # compiled into the library: nothing to do
    with pytest.raises(gfdm_amd.GfdmHipError):
        gfdm_amd.precompile(127, 16, 2)                                # generic family only
    assert os.stat(tmp_path).st_mode & 0o022 == 0                      # (pytest's tmp_path is private already)
    prev = gfdm_amd.set_jit(gfdm_amd.JIT_BACKGROUND)
    try:
        assert gfdm_amd.set_jit(99) == gfdm_amd.JIT_BACKGROUND and gfdm_amd.set_jit(gfdm_amd.JIT_OFF) == gfdm_amd.JIT_AUTO
        assert gfdm_amd.set_jit(True) == gfdm_amd.JIT_OFF and gfdm_amd.set_jit(prev) == gfdm_amd.JIT_IN_CONSTRUCTOR
    finally:
        gfdm_amd.set_jit(prev)
    mx = gfdm_amd.set_ic_matrix_cores(2)
    try:
        assert gfdm_amd.set_ic_matrix_cores(7) == 2 and gfdm_amd.set_ic_matrix_cores(0) == 2 and gfdm_amd.set_ic_matrix_cores(mx) == 0
    finally:
        gfdm_amd.set_ic_matrix_cores(mx)
    dx = gfdm_amd.set_dft_matrix_cores(2)                              # (default 1: the generic family's transforms on the matrix cores where faster)
    try:
        assert dx == 1 and gfdm_amd.set_dft_matrix_cores(9) == 2 and gfdm_amd.set_dft_matrix_cores(False) == 2 and gfdm_amd.set_dft_matrix_cores(dx) == 0
    finally:
        gfdm_amd.set_dft_matrix_cores(dx)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "gr-gfdm_amd", "python"), GFDM_HIP_CACHE_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "gfdm_amd.precompile", "6", "16", "2", "--parts", "rx", "9", "64", "2", "127", "16", "2"],
                       env=env, capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout.count("ready") == 2 and "generic kernel family" in r.stdout, r.stdout + r.stderr


def test_a_cache_directory_others_can_write_is_not_used(tmp_path, monkeypatch):
    """The code objects in the cache are handed to hipModuleLoadData: a directory that group / others can write (or that is a symbolic
    link) disables the disk cache -- the shape still compiles, in memory, and nothing is written there."""
    import gfdm_amd
    bad = tmp_path / "shared"
    bad.mkdir()
    os.chmod(bad, 0o777)
    monkeypatch.setenv("GFDM_HIP_CACHE_DIR", str(bad))
    L = gfdm_amd.lib()
    assert L.gfdm_hip_jit_build_for_testing(5, 16, 2, 3) == 0, L.gfdm_hip_last_error()
    assert os.listdir(bad) == []
    good = tmp_path / "private"
    link = tmp_path / "link"
    good.mkdir()
    os.symlink(good, link)
    monkeypatch.setenv("GFDM_HIP_CACHE_DIR", str(link))
    assert L.gfdm_hip_jit_build_for_testing(5, 16, 2, 3) == 0 and os.listdir(good) == []
    monkeypatch.setenv("GFDM_HIP_CACHE_DIR", str(good))
    assert L.gfdm_hip_jit_build_for_testing(5, 16, 2, 3) == 0 and len(os.listdir(good)) == 2


def test_gnuradio_block_wrappers_compile_against_a_mock_of_the_block_api():
    """gfdm/gr_blocks.h (complete gr::sync_block subclasses over the batched work() bodies) is compiled only where GNU Radio is
    installed, which is neither here nor on the GPU box.  Syntax-check it against tests/mock_gnuradio: a test double that declares the
    few names of GNU Radio's public block API the header uses (no reference code is built with it, nothing links against it)."""
    import subprocess
    inc = ["-I" + os.path.join(ROOT, "tests", "mock_gnuradio"), "-I" + os.path.join(ROOT, "gr-gfdm_amd", "cpp", "include"),
           "-I" + os.path.join(ROOT, "include")]
    src = os.path.join(ROOT, "gr-gfdm_amd", "cpp", "src", "gr_blocks.cc")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only"] + inc + [src])
    pre = subprocess.run(["g++", "-std=c++17", "-E"] + inc + [src], check=True, capture_output=True, text=True).stdout
    for cls in ("hip_simple_modulator_cc", "hip_simple_receiver_cc", "hip_advanced_receiver_sb_cc", "hip_transmitter_cc", "hip_channel_estimator_cc",
                "hip_resource_mapper_cc", "hip_cyclic_prefixer_cc"):
        assert "class " + cls in pre                      # the mock really switched the wrappers on
