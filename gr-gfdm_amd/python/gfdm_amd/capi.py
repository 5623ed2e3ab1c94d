"""ctypes binding of the C-ABI in include/gfdm_hip.h (gr-gfdm_amd/lib/libgfdm_hip.so).

Host-side mirror of the reference's kernel-class surface for Python callers that
work on whole batches and on device-resident torch tensors (bench.py, the
multi-GPU sharding helper).  The pybind11 module `gfdm_python` is the drop-in for
the reference's own binding; this module adds nothing numerically, it only
forwards pointers.  There is no CPU fallback: a missing library or GPU raises.
"""
import ctypes
import mmap
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.normpath(os.path.join(_PKG, "..", "..", "lib"))
LIB_PATH = os.environ.get("GFDM_HIP_LIB") or os.path.join(LIB_DIR, "libgfdm_hip.so")   # override: A/B builds of the kernels

OK, EINVAL_TAPS, EINVAL_OVERLAP, EINVAL, ENODEV, EHIP, ENOMEM, EUNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7
DECIDE = {"auto": -1, "nearest": 0, "qpsk": 1, "bpsk": 2}

_lib = None


class GfdmHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("gfdm_hip error %d: %s" % (status, message))
        self.status = status


def lib():
    """Load libgfdm_hip.so (built by `make -C gr-gfdm_amd` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `make -C gr-gfdm_amd` (no CPU fallback exists)" % LIB_PATH)
    # PyTorch-ROCm ships its own copy of the HIP runtime (torch/lib/libamdhip64.so).  A process can only work with ONE runtime:
    # if /opt/rocm's gets initialised first (through this library), torch later reports "No HIP GPUs are available".  The device
    # path of this package hands torch tensors to the library, so when torch is installed its runtime is loaded first and
    # libgfdm_hip.so binds to it (same SONAME); without torch the library uses /opt/rocm's.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, cp = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_char_p
    sig = {
        "gfdm_hip_strerror": (cp, [i32]),
        "gfdm_hip_last_error": (cp, []),
        "gfdm_hip_device_count": (i32, []),
        "gfdm_hip_force_generic_family_for_testing": (i32, [i32]),
        "gfdm_hip_set_jit": (i32, [i32]),
        "gfdm_hip_precompile": (i32, [i32, i32, i32, ctypes.c_uint]),
        "gfdm_hip_set_ic_matrix_cores": (i32, [i32]),
        "gfdm_hip_quiesce": (None, []),
        "gfdm_hip_set_dft_matrix_cores": (i32, [i32]),
        "gfdm_hip_jit_build_for_testing": (i32, [i32, i32, i32, i32]),
        "gfdm_hip_version": (cp, []),
        "gfdm_hip_build_id": (cp, []),
        "gfdm_hip_register_host": (i32, [vp, ctypes.c_size_t]),
        "gfdm_hip_unregister_host": (i32, [vp]),
        "gfdm_hip_set_host_pipeline": (i32, [i32, i64, i32, i32, i32]),
        "gfdm_hip_get_host_pipeline": (i32, [vp, vp, vp, vp, vp]),
        "gfdm_hip_host_call_stats": (i32, [vp, vp, vp, vp, vp, vp]),
        "gfdm_hip_host_call_times": (i32, [vp]),
        "gfdm_hip_set_host_streaming_copies_for_testing": (i32, [i32]),
        "gfdm_hip_modulator_create": (i32, [ctypes.POINTER(vp), i32, i32, i32, vp, i32, i32]),
        "gfdm_hip_modulator_destroy": (i32, [vp]),
        "gfdm_hip_modulator_block_size": (i32, [vp]),
        "gfdm_hip_modulator_filter_taps": (i32, [vp, vp]),
        "gfdm_hip_modulator_kernel_name": (cp, [vp]),
        "gfdm_hip_modulator_work_host": (i32, [vp, vp, vp, i64]),
        "gfdm_hip_modulator_work_device": (i32, [vp, vp, vp, i64, vp]),
        "gfdm_hip_receiver_create": (i32, [ctypes.POINTER(vp), i32, i32, i32, vp, i32, i32]),
        "gfdm_hip_receiver_destroy": (i32, [vp]),
        "gfdm_hip_receiver_block_size": (i32, [vp]),
        "gfdm_hip_receiver_timeslots": (i32, [vp]),
        "gfdm_hip_receiver_subcarriers": (i32, [vp]),
        "gfdm_hip_receiver_overlap": (i32, [vp]),
        "gfdm_hip_receiver_filter_taps": (i32, [vp, vp]),
        "gfdm_hip_receiver_ic_filter_taps": (i32, [vp, vp]),
        "gfdm_hip_receiver_kernel_name": (cp, [vp]),
        "gfdm_hip_receiver_demodulate_host": (i32, [vp, vp, vp, vp, i64]),
        "gfdm_hip_receiver_demodulate_device": (i32, [vp, vp, vp, vp, i64, vp]),
        "gfdm_hip_receiver_fft_filter_downsample_host": (i32, [vp, vp, vp, vp, i64]),
        "gfdm_hip_receiver_fft_filter_downsample_device": (i32, [vp, vp, vp, vp, i64, vp]),
        "gfdm_hip_receiver_transform_subcarriers_to_td_host": (i32, [vp, vp, vp, i64]),
        "gfdm_hip_receiver_transform_subcarriers_to_td_device": (i32, [vp, vp, vp, i64, vp]),
        "gfdm_hip_receiver_cancel_sc_interference_host": (i32, [vp, vp, vp, vp, i64]),
        "gfdm_hip_receiver_cancel_sc_interference_device": (i32, [vp, vp, vp, vp, i64, vp]),
        "gfdm_hip_advanced_receiver_create": (i32, [ctypes.POINTER(vp), i32, i32, i32, vp, i32, vp, i32, i32, vp, i32, i32, i32, i32]),
        "gfdm_hip_advanced_receiver_destroy": (i32, [vp]),
        "gfdm_hip_advanced_receiver_block_size": (i32, [vp]),
        "gfdm_hip_advanced_receiver_set_ic": (i32, [vp, i32]),
        "gfdm_hip_advanced_receiver_get_ic": (i32, [vp]),
        "gfdm_hip_advanced_receiver_set_phase_compensation": (i32, [vp, i32]),
        "gfdm_hip_advanced_receiver_get_phase_compensation": (i32, [vp]),
        "gfdm_hip_advanced_receiver_decision": (i32, [vp]),
        "gfdm_hip_advanced_receiver_kernel_name": (cp, [vp]),
        "gfdm_hip_advanced_receiver_work_host": (i32, [vp, vp, vp, vp, i64]),
        "gfdm_hip_advanced_receiver_work_device": (i32, [vp, vp, vp, vp, i64, vp]),
        "gfdm_hip_receiver_configure_frames": (i32, [vp, i32, i32, vp, i32, i32]),
        "gfdm_hip_receiver_demodulate_frames_host": (i32, [vp, vp, vp, vp, i32, i64]),
        "gfdm_hip_receiver_demodulate_frames_device": (i32, [vp, vp, vp, vp, i32, i64, vp]),
        "gfdm_hip_advanced_receiver_configure_frames": (i32, [vp, i32, i32, vp, i32, i32]),
        "gfdm_hip_advanced_receiver_work_frames_host": (i32, [vp, vp, vp, vp, i32, i64]),
        "gfdm_hip_advanced_receiver_work_frames_device": (i32, [vp, vp, vp, vp, i32, i64, vp]),
        "gfdm_hip_transmitter_create": (i32, [ctypes.POINTER(vp)] + [i32] * 6 + [vp, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32]),
        "gfdm_hip_transmitter_destroy": (i32, [vp]),
        "gfdm_hip_transmitter_input_vector_size": (i32, [vp]),
        "gfdm_hip_transmitter_output_vector_size": (i32, [vp]),
        "gfdm_hip_transmitter_block_size": (i32, [vp]),
        "gfdm_hip_transmitter_n_cyclic_shifts": (i32, [vp]),
        "gfdm_hip_transmitter_cyclic_shift": (i32, [vp, i32]),
        "gfdm_hip_transmitter_kernel_name": (cp, [vp]),
        "gfdm_hip_transmitter_work_host": (i32, [vp, vp, i32, vp, i32, i64]),
        "gfdm_hip_transmitter_work_device": (i32, [vp, vp, i32, vp, i32, i64, vp]),
        "gfdm_hip_transmitter_modulate_host": (i32, [vp, vp, vp, i32, i64]),
        "gfdm_hip_transmitter_modulate_device": (i32, [vp, vp, vp, i32, i64, vp]),
        "gfdm_hip_transmitter_add_frame_host": (i32, [vp, vp, vp, i32, i64]),
        "gfdm_hip_transmitter_add_frame_device": (i32, [vp, vp, vp, i32, i64, vp]),
        "gfdm_hip_resource_mapper_create": (i32, [ctypes.POINTER(vp), i32, i32, i32, vp, i32, i32, i32]),
        "gfdm_hip_resource_mapper_destroy": (i32, [vp]),
        "gfdm_hip_resource_mapper_block_size": (i32, [vp]),
        "gfdm_hip_resource_mapper_frame_size": (i32, [vp]),
        "gfdm_hip_resource_mapper_map_host": (i32, [vp, vp, vp, i32, i64]),
        "gfdm_hip_resource_mapper_map_device": (i32, [vp, vp, vp, i32, i64, vp]),
        "gfdm_hip_resource_mapper_demap_host": (i32, [vp, vp, vp, i32, i64]),
        "gfdm_hip_resource_mapper_demap_device": (i32, [vp, vp, vp, i32, i64, vp]),
        "gfdm_hip_cyclic_prefixer_create": (i32, [ctypes.POINTER(vp), i32, i32, i32, i32, vp, i32, i32, i32]),
        "gfdm_hip_cyclic_prefixer_destroy": (i32, [vp]),
        "gfdm_hip_cyclic_prefixer_block_size": (i32, [vp]),
        "gfdm_hip_cyclic_prefixer_frame_size": (i32, [vp]),
        "gfdm_hip_cyclic_prefixer_cyclic_shift": (i32, [vp]),
        "gfdm_hip_cyclic_prefixer_add_host": (i32, [vp, vp, vp, i32, i64]),
        "gfdm_hip_cyclic_prefixer_add_device": (i32, [vp, vp, vp, i32, i64, vp]),
        "gfdm_hip_cyclic_prefixer_remove_host": (i32, [vp, vp, vp, i64]),
        "gfdm_hip_cyclic_prefixer_remove_device": (i32, [vp, vp, vp, i64, vp]),
        "gfdm_hip_channel_estimator_create": (i32, [ctypes.POINTER(vp), i32, i32, i32, i32, i32, vp, i32, i32]),
        "gfdm_hip_channel_estimator_destroy": (i32, [vp]),
        "gfdm_hip_channel_estimator_timeslots": (i32, [vp]),
        "gfdm_hip_channel_estimator_fft_len": (i32, [vp]),
        "gfdm_hip_channel_estimator_active_subcarriers": (i32, [vp]),
        "gfdm_hip_channel_estimator_frame_len": (i32, [vp]),
        "gfdm_hip_channel_estimator_is_dc_free": (i32, [vp]),
        "gfdm_hip_channel_estimator_filtered_len": (i32, [vp]),
        "gfdm_hip_channel_estimator_kernel_name": (cp, [vp]),
        "gfdm_hip_channel_estimator_preamble_filter_taps": (i32, [vp, vp]),
        "gfdm_hip_channel_estimator_estimate_frame_host": (i32, [vp, vp, vp, i64]),
        "gfdm_hip_channel_estimator_estimate_frame_device": (i32, [vp, vp, vp, i64, vp]),
        "gfdm_hip_channel_estimator_estimate_preamble_channel_host": (i32, [vp, vp, vp, i64]),
        "gfdm_hip_channel_estimator_estimate_preamble_channel_device": (i32, [vp, vp, vp, i64, vp]),
        "gfdm_hip_channel_estimator_filter_preamble_estimate_host": (i32, [vp, vp, vp, i64]),
        "gfdm_hip_channel_estimator_filter_preamble_estimate_device": (i32, [vp, vp, vp, i64, vp]),
        "gfdm_hip_channel_estimator_interpolate_frame_host": (i32, [vp, vp, vp, i64]),
        "gfdm_hip_channel_estimator_interpolate_frame_device": (i32, [vp, vp, vp, i64, vp]),
        "gfdm_hip_channel_estimator_prepare_for_zf_host": (i32, [vp, vp, vp, i64]),
        "gfdm_hip_channel_estimator_prepare_for_zf_device": (i32, [vp, vp, vp, i64, vp]),
        "gfdm_hip_channel_estimator_estimate_snr_host": (i32, [vp, vp, vp, vp, i64]),
        "gfdm_hip_channel_estimator_estimate_snr_device": (i32, [vp, vp, vp, vp, i64, vp]),
        "gfdm_hip_receiver_set_channel_estimator": (i32, [vp, vp]),
        "gfdm_hip_advanced_receiver_set_channel_estimator": (i32, [vp, vp]),
        "gfdm_hip_receiver_io_layout": (i32, [vp, i32, i32, vp, vp, vp]),
        "gfdm_hip_advanced_receiver_io_layout": (i32, [vp, i32, i32, vp, vp, vp]),
        "gfdm_hip_receiver_demodulate_estimated_host": (i32, [vp, vp, vp, vp, i32, i32, i64]),
        "gfdm_hip_receiver_demodulate_estimated_device": (i32, [vp, vp, vp, vp, i32, i32, i64, vp]),
        "gfdm_hip_advanced_receiver_work_estimated_host": (i32, [vp, vp, vp, vp, i32, i32, i64]),
        "gfdm_hip_advanced_receiver_work_estimated_device": (i32, [vp, vp, vp, vp, i32, i32, i64, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError here == the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    L._gfdm_symbols = sorted(sig)
    # background builds must not outlive the interpreter's tear-down of the HIP runtime (gfdm_hip_quiesce: include/gfdm_hip.h)
    import atexit
    atexit.register(L.gfdm_hip_quiesce)
    _lib = L
    return L


def exported_symbols():
    return list(lib()._gfdm_symbols)


def _check(status):
    if status == OK:
        return
    L = lib()
    msg = (L.gfdm_hip_last_error() or b"").decode() or L.gfdm_hip_strerror(status).decode()
    if status in (EINVAL_TAPS, EINVAL_OVERLAP, EINVAL):
        raise ValueError(msg)            # std::invalid_argument in the reference (pybind11 maps it to ValueError)
    raise GfdmHipError(status, msg)


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


def _is_tensor(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


def _dev_ptr(t, n_elems, what, device=None):
    import torch
    if not t.is_cuda or t.dtype != torch.complex64 or not t.is_contiguous():
        raise TypeError("%s must be a contiguous complex64 CUDA/HIP tensor" % what)
    if device is not None and t.device.index != device:
        raise RuntimeError("%s lives on GPU %d, the kernel handle on GPU %d" % (what, t.device.index, device))
    if t.numel() != n_elems:
        raise RuntimeError("%s has %d elements, expected %d" % (what, t.numel(), n_elems))
    return t.data_ptr()


def _stream_ptr(stream, device=None):
    if stream is None:
        import torch
        return torch.cuda.current_stream(device).cuda_stream      # the current stream of the HANDLE's GPU, not of torch's current one
    if hasattr(stream, "cuda_stream"):
        return stream.cuda_stream
    return int(stream)


class generic_family_for_testing:
    """TEST HOOK: `with generic_family_for_testing(): h = Demodulator(...)` creates h on the generic kernel family even where the
    tuned row-lane family serves the shape (gfdm_hip_force_generic_family_for_testing); used by tests/ to run both families."""

    def __enter__(self):
        self._prev = lib().gfdm_hip_force_generic_family_for_testing(1)
        return self

    def __exit__(self, *exc):
        lib().gfdm_hip_force_generic_family_for_testing(self._prev)
        return False


JIT_OFF, JIT_IN_CONSTRUCTOR, JIT_BACKGROUND, JIT_AUTO = 0, 1, 2, 3


def set_jit(mode):
    """gfdm_hip_set_jit: when the row-lane kernels of a shape outside the compiled list are instantiated (hiprtc) -- JIT_OFF never (generic
    family), JIT_IN_CONSTRUCTOR, JIT_BACKGROUND (the handle starts on the generic family and switches over), JIT_AUTO (default: in the
    constructor when cached or quick, else in the background).  True / False mean JIT_IN_CONSTRUCTOR / JIT_OFF.  Returns the previous mode."""
    if mode is True:
        mode = JIT_IN_CONSTRUCTOR
    elif mode is False:
        mode = JIT_OFF
    return lib().gfdm_hip_set_jit(int(mode))


def precompile(timeslots, subcarriers, overlap, parts=0):
    """gfdm_hip_precompile: compile the tuned kernels of a shape into the disk cache without creating a handle (no GPU needed).
    parts: bit mask (receive 1, receive + IC 2, preamble-equalised receive 4, modulate 8, estimator 16), 0 = all."""
    _check(lib().gfdm_hip_precompile(int(timeslots), int(subcarriers), int(overlap), int(parts)))


def set_ic_matrix_cores(mode):
    """gfdm_hip_set_ic_matrix_cores: where handles created afterwards run the interference-cancellation rounds -- 0 / False vector ALU
    only, 1 / True (default) matrix cores where they are the faster form (subcarriers >= 128), 2 matrix cores wherever the form applies.
    Returns the previous mode."""
    return lib().gfdm_hip_set_ic_matrix_cores(int(mode))


def quiesce():
    """gfdm_hip_quiesce: drop the queued background builds of the run-time instantiated kernels, wait for the ones in flight (they finish
    their compile without touching the GPU).  Registered with atexit when the library is loaded."""
    lib().gfdm_hip_quiesce()


def build_id():
    """gfdm_hip_build_id: hash of the sources and flags the loaded library was built from."""
    return lib().gfdm_hip_build_id().decode()


HOST_ZERO_COPY, HOST_COPY_ENGINES = 0, 1


def set_host_pipeline(mode=-1, chunk_bytes=-1, depth=-1, copy_threads=-1, kernel_streams=-1):
    """gfdm_hip_set_host_pipeline: tuning of the *_host calls' bounce through pinned staging (negative / omitted = unchanged).  Returns the
    previous (mode, chunk_bytes, depth, copy_threads, kernel_streams)."""
    prev = get_host_pipeline()
    _check(lib().gfdm_hip_set_host_pipeline(int(mode), int(chunk_bytes), int(depth), int(copy_threads), int(kernel_streams)))
    return prev


def get_host_pipeline():
    m, c, d, t, k = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _check(lib().gfdm_hip_get_host_pipeline(ctypes.byref(m), ctypes.byref(c), ctypes.byref(d), ctypes.byref(t), ctypes.byref(k)))
    return m.value, c.value, d.value, t.value, k.value


def host_call_stats():
    """What the calling thread's last *_host call did: dict(chunks, chunk_blocks, staged_bytes, direct_mask, mode, copy_threads)."""
    ch, cb, sb = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    dm, md, ct = ctypes.c_uint(0), ctypes.c_int(0), ctypes.c_int(0)
    _check(lib().gfdm_hip_host_call_stats(ctypes.byref(ch), ctypes.byref(cb), ctypes.byref(sb), ctypes.byref(dm), ctypes.byref(md), ctypes.byref(ct)))
    ns = (ctypes.c_int64 * 5)()
    _check(lib().gfdm_hip_host_call_times(ns))
    return dict(chunks=ch.value, chunk_blocks=cb.value, staged_bytes=sb.value, direct_mask=dm.value, mode=md.value, copy_threads=ct.value,
                ns=dict(setup=ns[0], copy=ns[1], launch=ns[2], post=ns[3], wait=ns[4]))


class registered_host:
    """gfdm_hip_register_host as a context manager: `with registered_host(a, b): dem.demodulate(a, out=b)` -- the numpy arrays are pinned and
    mapped for the GPUs, *_host calls on them run in place (no bounce).  A long-lived buffer is registered once with register_host()."""

    def __init__(self, *arrays):
        self._arrays = arrays

    def __enter__(self):
        done = []
        try:
            for a in self._arrays:
                register_host(a)
                done.append(a)
        except Exception:
            for a in done:
                unregister_host(a)
            raise
        return self

    def __exit__(self, *exc):
        for a in self._arrays:
            unregister_host(a)
        return False


PAGE = mmap.PAGESIZE


def aligned_empty(shape, dtype=np.complex64):
    """An uninitialised C-contiguous array that starts on a page boundary and owns its pages to the end of the last one: what register_host takes
    (gfdm_hip_register_host pins whole pages).  The backing buffer stays alive with the array."""
    import mmap
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    size = max(PAGE, -(-n // PAGE) * PAGE)
    buf = mmap.mmap(-1, size)                          # anonymous, page aligned, page granular
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def aligned_copy(array):
    out = aligned_empty(array.shape, array.dtype)
    out[...] = array
    return out


def register_host(array):
    """gfdm_hip_register_host on an array from aligned_empty / aligned_copy (or any array that starts on a page boundary and owns the rest of its last page)."""
    if array.ctypes.data % PAGE:
        raise ValueError("register_host: the array must start on a page boundary (gfdm_amd.aligned_empty / aligned_copy)")
    _check(lib().gfdm_hip_register_host(array.ctypes.data, -(-array.nbytes // PAGE) * PAGE))


def unregister_host(array):
    _check(lib().gfdm_hip_unregister_host(array.ctypes.data))


def set_dft_matrix_cores(mode):
    """gfdm_hip_set_dft_matrix_cores: where the generic kernel family of handles created afterwards runs its timeslot transforms on the
    matrix cores (f32 MFMA; from 32 timeslots on) -- 0 / False never, 1 / True (default) where that is the faster form, 2 wherever the form
    fits.  Returns the previous mode."""
    return lib().gfdm_hip_set_dft_matrix_cores(int(mode))


class _Kernel:
    """Shared plumbing: host (numpy) and device (torch) batched calls."""
    _destroy = None
    _frames_prefix = None          # C-ABI name stem of the *_frames_* entry points (receivers only)
    _configure_frames = None
    _io_layout = None
    _dev = 0                       # GPU the handle was created on: every tensor and the stream of a call must belong to it

    def _dp(self, t, n_elems, what):
        return _dev_ptr(t, n_elems, what, self._dev)

    def _sp(self, stream):
        return _stream_ptr(stream, self._dev)

    def _layout(self, estimated, nout_arg):
        """(n_in, n_out, estimator fft_len) of one frames / estimated call, asked from the library (gfdm_hip_*_io_layout): it alone
        knows what its kernels write for this configuration, so no output buffer is ever sized from Python-side bookkeeping."""
        n_in, n_out, fft_len = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _check(getattr(lib(), self._io_layout)(self._h, int(estimated), int(nout_arg), ctypes.byref(n_in), ctypes.byref(n_out), ctypes.byref(fft_len)))
        return n_in.value, n_out.value, fft_len.value

    # ---- raw frames in, demapped symbols out ----
    def configure_frames(self, frame_len, cp_len, subcarrier_map=None, per_timeslot=True):
        """Declare the frame layout once: frames of frame_len samples whose block starts cp_len samples in; with a
        subcarrier_map only the active subcarriers' symbols are emitted, in resource-mapper order."""
        smap = np.ascontiguousarray([] if subcarrier_map is None else subcarrier_map, dtype=np.int32)
        fn = getattr(lib(), self._configure_frames)
        _check(fn(self._h, int(frame_len), int(cp_len), smap.ctypes.data if smap.size else None, smap.size, int(bool(per_timeslot))))

    def demodulate_frames(self, frames, f_eq=None, noutput_size=None, out=None, stream=None):
        """frames: nframes * frame_len samples; returns (nframes, noutput_size) symbols (all active symbols by default)."""
        L = lib()
        nout_arg = -1 if noutput_size is None else int(noutput_size)
        frame_len, nout, _ = self._layout(False, nout_arg)
        N = self.block_size()
        if _is_tensor(frames):
            import torch
            if frames.numel() % frame_len:
                raise RuntimeError("frames size(%d) MUST be a multiple of frame_len(%d)!" % (frames.numel(), frame_len))
            nb = frames.numel() // frame_len
            out = torch.empty(nb, nout, dtype=torch.complex64, device=frames.device) if out is None else out
            feq = None if f_eq is None else self._dp(f_eq, nb * N, "f_eq")
            _check(getattr(L, self._frames_prefix + "_device")(self._h, self._dp(out, nb * nout, "out"), self._dp(frames, nb * frame_len, "frames"),
                                                              feq, nout_arg, nb, self._sp(stream)))
            return out
        x = _c64(frames)
        if x.size % frame_len:
            raise RuntimeError("frames size(%d) MUST be a multiple of frame_len(%d)!" % (x.size, frame_len))
        nb = x.size // frame_len
        e = None if f_eq is None else _c64(f_eq)
        if e is not None and e.size != nb * N:
            raise RuntimeError("Channel vector size(%d) MUST be equal to nframes * block_size(%d)!" % (e.size, nb * N))
        res = np.empty((nb, nout), np.complex64)
        _check(getattr(L, self._frames_prefix + "_host")(self._h, res.ctypes.data, x.ctypes.data, None if e is None else e.ctypes.data, nout_arg, nb))
        return res

    # ---- channel estimator fused in front of the receiver ----
    def set_channel_estimator(self, estimator):
        """Attach a ChannelEstimator (or None): demodulate_estimated then derives the equaliser from each block's received preamble."""
        _check(getattr(lib(), self._set_estimator)(self._h, None if estimator is None else estimator._h))
        self._estimator = estimator                    # keeps the handle alive

    def demodulate_estimated(self, x, rx_preamble, preamble_stride=0, noutput_size=None, out=None, stream=None):
        """x: blocks (frames when configure_frames was called); rx_preamble: the received core preamble of every block, preamble b at
        b * preamble_stride (0 = packed, 2 * subcarriers).  Equals estimate_frame + demodulate_equalize, in one kernel."""
        L = lib()
        if getattr(self, "_estimator", None) is None:
            raise ValueError("set_channel_estimator has not been called on this handle")
        nout_arg = -1 if noutput_size is None else int(noutput_size)
        n_in, nout, fft_len = self._layout(True, nout_arg)
        stride = int(preamble_stride) if preamble_stride else 2 * fft_len
        if _is_tensor(x):
            import torch
            if x.numel() % n_in:
                raise RuntimeError("Input size(%d) MUST be a multiple of %d!" % (x.numel(), n_in))
            nb = x.numel() // n_in
            out = torch.empty(nb, nout, dtype=torch.complex64, device=x.device) if out is None else out
            need = (nb - 1) * stride + 2 * fft_len if nb else 0
            if rx_preamble.numel() < need:
                raise RuntimeError("rx_preamble has %d elements, at least %d are needed" % (rx_preamble.numel(), need))
            self._dp(rx_preamble, rx_preamble.numel(), "rx_preamble")
            _check(getattr(L, self._estimated_prefix + "_device")(self._h, self._dp(out, nb * nout, "out"), self._dp(x, nb * n_in, "in"),
                                                                 rx_preamble.data_ptr(), int(preamble_stride), nout_arg, nb, self._sp(stream)))
            return out
        a = _c64(x)
        if a.size % n_in:
            raise RuntimeError("Input size(%d) MUST be a multiple of %d!" % (a.size, n_in))
        nb = a.size // n_in
        pre = _c64(rx_preamble)
        need = (nb - 1) * stride + 2 * fft_len if nb else 0
        if pre.size < need:
            raise RuntimeError("rx_preamble size(%d) MUST be at least %d!" % (pre.size, need))
        res = np.empty((nb, nout), np.complex64)
        _check(getattr(L, self._estimated_prefix + "_host")(self._h, res.ctypes.data, a.ctypes.data, pre.ctypes.data, int(preamble_stride), nout_arg, nb))
        return res

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None and self._destroy:
            getattr(_lib, self._destroy)(h)
            self._h = None

    def _nblocks(self, size, what="Input"):
        n = self.block_size()
        if size % n:
            raise RuntimeError("%s vector size(%d) MUST be a multiple of block_size(%d)!" % (what, size, n))
        return size // n

    def _host(self, fn, x, extra=(), extra_ok_none=False, out=None):
        """numpy path: the *_host entry point on the arrays' memory.  `out`: an existing complex64 array to write into (e.g. one registered with
        register_host, so that the call runs in place on it)."""
        x = _c64(x)
        nb = self._nblocks(x.size)
        ptrs = []
        keep = []
        for e in extra:
            if e is None:
                if not extra_ok_none:
                    raise RuntimeError("missing input vector")
                ptrs.append(None)
            else:
                e = _c64(e)
                if e.size != x.size:
                    raise RuntimeError("Channel vector size(%d) MUST be equal to input size(%d)!" % (e.size, x.size))
                keep.append(e)
                ptrs.append(e.ctypes.data)
        if out is None:
            out = np.empty(x.shape, np.complex64)
        elif not (isinstance(out, np.ndarray) and out.dtype == np.complex64 and out.flags.c_contiguous and out.size == x.size):
            raise TypeError("out must be a C-contiguous complex64 numpy array of the input's size")
        _check(fn(self._h, out.ctypes.data, x.ctypes.data, *ptrs, nb))
        return out

    def _device(self, fn, out, x, extra=(), stream=None):
        n = x.numel()
        nb = self._nblocks(n)
        ptrs = [None if e is None else self._dp(e, n, "extra input") for e in extra]
        _check(fn(self._h, self._dp(out, n, "out"), self._dp(x, n, "in"), *ptrs, nb, self._sp(stream)))
        return out


def _taps_arg(taps):
    t = _c64(np.asarray(taps).ravel())
    return t, t.ctypes.data, t.size


class Modulator(_Kernel):
    """gr::gfdm::modulator_kernel_cc (include/gfdm/modulator_kernel_cc.h:41-51) on the GPU."""
    _destroy = "gfdm_hip_modulator_destroy"

    def __init__(self, timeslots, subcarriers, overlap, taps, device=0):
        L = lib()
        t, tp, tn = _taps_arg(taps)
        h = ctypes.c_void_p()
        _check(L.gfdm_hip_modulator_create(ctypes.byref(h), timeslots, subcarriers, overlap, tp, tn, device))
        self._h = h
        self._dev = int(device)
        self._n = timeslots * subcarriers
        self._ntaps = tn

    def block_size(self):
        return self._n

    def kernel_name(self):
        return lib().gfdm_hip_modulator_kernel_name(self._h).decode()

    def filter_taps(self):
        out = np.empty(self._ntaps, np.complex64)
        _check(lib().gfdm_hip_modulator_filter_taps(self._h, out.ctypes.data))
        return out

    def modulate(self, x, out=None, stream=None):
        """numpy in -> numpy out (any whole number of blocks); torch CUDA tensor in -> `out` tensor, async on `stream`."""
        if _is_tensor(x):
            import torch
            out = torch.empty_like(x) if out is None else out
            return self._device(lib().gfdm_hip_modulator_work_device, out, x, stream=stream)
        return self._host(lib().gfdm_hip_modulator_work_host, x, out=out)


class Demodulator(_Kernel):
    """gr::gfdm::receiver_kernel_cc (include/gfdm/receiver_kernel_cc.h:52-89) on the GPU."""
    _destroy = "gfdm_hip_receiver_destroy"
    _frames_prefix = "gfdm_hip_receiver_demodulate_frames"
    _configure_frames = "gfdm_hip_receiver_configure_frames"
    _io_layout = "gfdm_hip_receiver_io_layout"
    _estimated_prefix = "gfdm_hip_receiver_demodulate_estimated"
    _set_estimator = "gfdm_hip_receiver_set_channel_estimator"

    def __init__(self, timeslots, subcarriers, overlap, taps, device=0):
        L = lib()
        t, tp, tn = _taps_arg(taps)
        h = ctypes.c_void_p()
        _check(L.gfdm_hip_receiver_create(ctypes.byref(h), timeslots, subcarriers, overlap, tp, tn, device))
        self._h = h
        self._dev = int(device)
        self._M, self._K, self._L = timeslots, subcarriers, overlap

    def timeslots(self):
        return lib().gfdm_hip_receiver_timeslots(self._h)

    def subcarriers(self):
        return lib().gfdm_hip_receiver_subcarriers(self._h)

    def overlap(self):
        return lib().gfdm_hip_receiver_overlap(self._h)

    def block_size(self):
        return self._M * self._K

    def kernel_name(self):
        return lib().gfdm_hip_receiver_kernel_name(self._h).decode()

    def filter_taps(self):
        out = np.empty(self._M * self._L, np.complex64)
        _check(lib().gfdm_hip_receiver_filter_taps(self._h, out.ctypes.data))
        return out

    def ic_filter_taps(self):
        out = np.empty(self._M, np.complex64)
        _check(lib().gfdm_hip_receiver_ic_filter_taps(self._h, out.ctypes.data))
        return out

    def _call(self, name, x, extra, out, stream, extra_ok_none=False):
        L = lib()
        if _is_tensor(x):
            import torch
            out = torch.empty_like(x) if out is None else out
            return self._device(getattr(L, name + "_device"), out, x, extra, stream)
        return self._host(getattr(L, name + "_host"), x, extra, extra_ok_none, out=out)

    def demodulate(self, x, out=None, stream=None):
        return self._call("gfdm_hip_receiver_demodulate", x, (None,), out, stream, True)

    def demodulate_equalize(self, x, f_eq, out=None, stream=None):
        return self._call("gfdm_hip_receiver_demodulate", x, (f_eq,), out, stream)

    def fft_filter_downsample(self, x, out=None, stream=None):
        return self._call("gfdm_hip_receiver_fft_filter_downsample", x, (None,), out, stream, True)

    def fft_equalize_filter_downsample(self, x, f_eq, out=None, stream=None):
        return self._call("gfdm_hip_receiver_fft_filter_downsample", x, (f_eq,), out, stream)

    def transform_subcarriers_to_td(self, x, out=None, stream=None):
        return self._call("gfdm_hip_receiver_transform_subcarriers_to_td", x, (), out, stream)

    def cancel_sc_interference(self, td, fd, out=None, stream=None):
        return self._call("gfdm_hip_receiver_cancel_sc_interference", td, (fd,), out, stream)


class AdvancedReceiver(_Kernel):
    """gr::gfdm::advanced_receiver_kernel_cc (include/gfdm/advanced_receiver_kernel_cc.h:37-78) on the GPU.

    The reference takes a gr::digital::constellation_sptr; here the constellation is its points() array plus a
    decision rule ('auto' picks the QPSK/BPSK sign tests when the points are those constellations).  Only GNU Radio's UNIT constellations
    -- (+-1 +-j)/sqrt 2 in the order --, +-, -+, ++ and -1, +1, every component within four float32 ulps -- take the sign tests (and the
    matrix-core cancellation rounds); scaled or rotated points are decided by the nearest-point rule over the points as given, also when
    'qpsk' / 'bpsk' was asked for.  decision_rule() reports the rule the handle runs."""
    _destroy = "gfdm_hip_advanced_receiver_destroy"
    _frames_prefix = "gfdm_hip_advanced_receiver_work_frames"
    _configure_frames = "gfdm_hip_advanced_receiver_configure_frames"
    _io_layout = "gfdm_hip_advanced_receiver_io_layout"
    _estimated_prefix = "gfdm_hip_advanced_receiver_work_estimated"
    _set_estimator = "gfdm_hip_advanced_receiver_set_channel_estimator"

    def __init__(self, timeslots, subcarriers, overlap, taps, subcarrier_map, ic_iter, constellation_points,
                 do_phase_compensation=0, decision="auto", device=0):
        L = lib()
        t, tp, tn = _taps_arg(taps)
        smap = np.ascontiguousarray(subcarrier_map, dtype=np.int32)
        pts = _c64(np.asarray(constellation_points).ravel())
        h = ctypes.c_void_p()
        _check(L.gfdm_hip_advanced_receiver_create(ctypes.byref(h), timeslots, subcarriers, overlap, tp, tn,
                                                   smap.ctypes.data, smap.size, ic_iter, pts.ctypes.data, pts.size,
                                                   DECIDE[decision], do_phase_compensation, device))
        self._h = h
        self._dev = int(device)
        self._n = timeslots * subcarriers

    def block_size(self):
        return self._n

    def kernel_name(self):
        return lib().gfdm_hip_advanced_receiver_kernel_name(self._h).decode()

    def decision_rule(self):
        """'nearest' | 'qpsk' | 'bpsk': the rule this handle runs (gfdm_hip_advanced_receiver_decision)"""
        code = lib().gfdm_hip_advanced_receiver_decision(self._h)
        return {v: k for k, v in DECIDE.items() if k != "auto"}[code]

    def set_ic(self, ic_iter):
        _check(lib().gfdm_hip_advanced_receiver_set_ic(self._h, ic_iter))

    def get_ic(self):
        return lib().gfdm_hip_advanced_receiver_get_ic(self._h)

    def set_phase_compensation(self, enable):
        _check(lib().gfdm_hip_advanced_receiver_set_phase_compensation(self._h, enable))

    def get_phase_compensation(self):
        return lib().gfdm_hip_advanced_receiver_get_phase_compensation(self._h)

    def _call(self, x, f_eq, out, stream):
        L = lib()
        if _is_tensor(x):
            import torch
            out = torch.empty_like(x) if out is None else out
            return self._device(L.gfdm_hip_advanced_receiver_work_device, out, x, (f_eq,), stream)
        return self._host(L.gfdm_hip_advanced_receiver_work_host, x, (f_eq,), True, out=out)

    def demodulate(self, x, out=None, stream=None):
        """generic_work: IC receiver without equaliser."""
        return self._call(x, None, out, stream)

    def demodulate_equalize(self, x, f_eq, out=None, stream=None):
        """generic_work_equalize: one f_eq vector per block."""
        return self._call(x, f_eq, out, stream)


class Transmitter(_Kernel):
    """gr::gfdm::transmitter_kernel (include/gfdm/transmitter_kernel.h:43-85) on the GPU: resource mapper -> modulator ->
    cyclic prefix/suffix with cyclic shift + window ramp -> preamble as ONE kernel, every cyclic shift ("port") at once."""
    _destroy = "gfdm_hip_transmitter_destroy"

    def __init__(self, timeslots, subcarriers, active_subcarriers, cp_len, cs_len, ramp_len, subcarrier_map, per_timeslot, overlap,
                 frequency_taps, window_taps, cyclic_shifts, preambles, device=0):
        L = lib()
        t, tp, tn = _taps_arg(frequency_taps)
        w = _c64(np.asarray(window_taps).ravel())
        smap = np.ascontiguousarray(subcarrier_map, dtype=np.int32)
        shifts = np.ascontiguousarray(cyclic_shifts, dtype=np.int32)
        pre = [np.asarray(p).ravel() for p in preambles]
        if len(pre) != shifts.size:
            raise ValueError("Number of cyclic shifts and number of preambles do not match!")
        if any(p.size != pre[0].size for p in pre):
            raise ValueError("All preambles must have equal size!")
        pre = _c64(np.stack(pre)) if pre else np.zeros((0, 0), np.complex64)
        h = ctypes.c_void_p()
        _check(L.gfdm_hip_transmitter_create(ctypes.byref(h), timeslots, subcarriers, active_subcarriers, cp_len, cs_len, ramp_len,
                                             smap.ctypes.data, smap.size, int(bool(per_timeslot)), overlap, tp, tn, w.ctypes.data, w.size,
                                             shifts.ctypes.data, shifts.size, pre.ctypes.data, pre.shape[1] if pre.size else 0, device))
        self._h = h
        self._dev = int(device)
        self._shifts = [int(x) for x in shifts]

    def input_vector_size(self):
        return lib().gfdm_hip_transmitter_input_vector_size(self._h)

    def output_vector_size(self):
        return lib().gfdm_hip_transmitter_output_vector_size(self._h)

    def block_size(self):
        return lib().gfdm_hip_transmitter_block_size(self._h)

    def cyclic_shifts(self):
        return list(self._shifts)

    def kernel_name(self):
        return lib().gfdm_hip_transmitter_kernel_name(self._h).decode()

    def _split(self, x, ninput_size):
        n = self.input_vector_size() if ninput_size is None else int(ninput_size)
        size = x.numel() if _is_tensor(x) else np.asarray(x).size
        if n <= 0 or size % n:
            raise RuntimeError("input size(%d) MUST be a multiple of ninput_size(%d)!" % (size, n))
        return n, size // n

    def transmit(self, symbols, ninput_size=None, n_ports=None, stream=None):
        """All ports (or the first n_ports): list of (nblocks, output_vector_size) arrays / tensors, one per cyclic shift."""
        L = lib()
        n, nb = self._split(symbols, ninput_size)
        ports = len(self._shifts) if n_ports is None else n_ports
        F = self.output_vector_size()
        if _is_tensor(symbols):
            import torch
            outs = [torch.empty(nb, F, dtype=torch.complex64, device=symbols.device) for _ in range(ports)]
            arr = (ctypes.c_void_p * ports)(*[o.data_ptr() for o in outs])
            _check(L.gfdm_hip_transmitter_work_device(self._h, arr, ports, self._dp(symbols, nb * n, "in"), n, nb, self._sp(stream)))
            return outs
        x = _c64(symbols)
        outs = [np.empty((nb, F), np.complex64) for _ in range(ports)]
        arr = (ctypes.c_void_p * ports)(*[o.ctypes.data for o in outs])
        _check(L.gfdm_hip_transmitter_work_host(self._h, arr, ports, x.ctypes.data, n, nb))
        return outs

    def generic_work(self, symbols, ninput_size=None):
        """transmitter_kernel::generic_work: the frame of cyclic_shifts[0]."""
        return self.transmit(symbols, ninput_size, 1)[0]

    def modulate(self, symbols, ninput_size=None, stream=None):
        L = lib()
        n, nb = self._split(symbols, ninput_size)
        N = self.block_size()
        if _is_tensor(symbols):
            import torch
            out = torch.empty(nb, N, dtype=torch.complex64, device=symbols.device)
            _check(L.gfdm_hip_transmitter_modulate_device(self._h, out.data_ptr(), self._dp(symbols, nb * n, "in"), n, nb, self._sp(stream)))
            return out
        x = _c64(symbols)
        out = np.empty((nb, N), np.complex64)
        _check(L.gfdm_hip_transmitter_modulate_host(self._h, out.ctypes.data, x.ctypes.data, n, nb))
        return out

    def add_frame(self, blocks, cyclic_shift, stream=None):
        L = lib()
        N, F = self.block_size(), self.output_vector_size()
        if _is_tensor(blocks):
            import torch
            nb = blocks.numel() // N
            out = torch.empty(nb, F, dtype=torch.complex64, device=blocks.device)
            _check(L.gfdm_hip_transmitter_add_frame_device(self._h, out.data_ptr(), self._dp(blocks, nb * N, "in"), int(cyclic_shift), nb, self._sp(stream)))
            return out
        x = _c64(blocks)
        nb = x.size // N
        out = np.empty((nb, F), np.complex64)
        _check(L.gfdm_hip_transmitter_add_frame_host(self._h, out.ctypes.data, x.ctypes.data, int(cyclic_shift), nb))
        return out


class ResourceMapper(_Kernel):
    """gr::gfdm::resource_mapper_kernel_cc (include/gfdm/resource_mapper_kernel_cc.h:38-60; pybind name Resource_mapper): data symbols
    <-> the [subcarriers][timeslots] grid, stand-alone (Transmitter and the receivers' frame interface have it fused).  One block or a
    batch ([nblocks][size]); numpy in -> numpy out (host path), torch in -> torch out (device path, current stream unless given)."""
    _destroy = "gfdm_hip_resource_mapper_destroy"

    def __init__(self, timeslots, subcarriers, active_subcarriers, subcarrier_map, per_timeslot=True, device=0):
        smap = np.ascontiguousarray(subcarrier_map, dtype=np.int32)
        h = ctypes.c_void_p()
        _check(lib().gfdm_hip_resource_mapper_create(ctypes.byref(h), timeslots, subcarriers, active_subcarriers, smap.ctypes.data, smap.size,
                                                     int(bool(per_timeslot)), device))
        self._h = h
        self._dev = int(device)

    def block_size(self):
        return lib().gfdm_hip_resource_mapper_block_size(self._h)

    def frame_size(self):
        return lib().gfdm_hip_resource_mapper_frame_size(self._h)

    def _run(self, host, dev, x, n_in, n_out, size_arg, stream, out=None):
        size = x.numel() if _is_tensor(x) else np.asarray(x).size
        if n_in <= 0 or size % n_in:
            raise RuntimeError("input size(%d) MUST be a multiple of %d!" % (size, n_in))
        nb = size // n_in
        if _is_tensor(x):
            import torch
            if out is None:
                out = torch.empty(nb, n_out, dtype=torch.complex64, device=x.device)
            _check(dev(self._h, self._dp(out, nb * n_out, "out"), self._dp(x, size, "in"), size_arg, nb, self._sp(stream)))
            return out
        a = _c64(x)
        out = np.empty((nb, n_out), np.complex64)
        _check(host(self._h, out.ctypes.data, a.ctypes.data, size_arg, nb))
        return out[0] if (np.asarray(x).ndim == 1 and nb == 1) else out

    def map_to_resources(self, symbols, ninput_size=None, stream=None, out=None):
        """ninput_size symbols per block (default block_size) -> frame_size grid values per block, unfilled slots zero."""
        n = self.block_size() if ninput_size is None else int(ninput_size)
        if n == 0:
            raise RuntimeError("ninput_size 0: give the number of blocks through a [nblocks][0] input of the C-ABI instead")
        L = lib()
        return self._run(L.gfdm_hip_resource_mapper_map_host, L.gfdm_hip_resource_mapper_map_device, symbols, n, self.frame_size(), n, stream, out)

    def demap_from_resources(self, grid, noutput_size=None, stream=None, out=None):
        n = self.block_size() if noutput_size is None else int(noutput_size)
        L = lib()
        return self._run(L.gfdm_hip_resource_mapper_demap_host, L.gfdm_hip_resource_mapper_demap_device, grid, self.frame_size(), n, n, stream, out)


class CyclicPrefixer(_Kernel):
    """gr::gfdm::add_cyclic_prefix_cc (include/gfdm/add_cyclic_prefix_cc.h:40-60; pybind name Cyclic_prefixer): cyclic prefix / suffix
    with cyclic shift + window ramps, and prefix removal, stand-alone.  One block or a batch; numpy -> numpy, torch -> torch."""
    _destroy = "gfdm_hip_cyclic_prefixer_destroy"

    def __init__(self, block_len, cp_len, cs_len, ramp_len, window_taps, cyclic_shift=0, device=0):
        w = _c64(np.asarray(window_taps).ravel())
        h = ctypes.c_void_p()
        _check(lib().gfdm_hip_cyclic_prefixer_create(ctypes.byref(h), block_len, cp_len, cs_len, ramp_len, w.ctypes.data, w.size, int(cyclic_shift),
                                                     device))
        self._h = h
        self._dev = int(device)

    def block_size(self):
        return lib().gfdm_hip_cyclic_prefixer_block_size(self._h)

    def frame_size(self):
        return lib().gfdm_hip_cyclic_prefixer_frame_size(self._h)

    def cyclic_shift(self):
        return lib().gfdm_hip_cyclic_prefixer_cyclic_shift(self._h)

    def _batch(self, x, n_in):
        size = x.numel() if _is_tensor(x) else np.asarray(x).size
        if size % n_in:
            raise RuntimeError("input size(%d) MUST be a multiple of %d!" % (size, n_in))
        return size, size // n_in

    def add_cyclic_prefix(self, blocks, cyclic_shift=None, stream=None, out=None):
        L = lib()
        s = self.cyclic_shift() if cyclic_shift is None else int(cyclic_shift)
        N, F = self.block_size(), self.frame_size()
        size, nb = self._batch(blocks, N)
        if _is_tensor(blocks):
            import torch
            if out is None:
                out = torch.empty(nb, F, dtype=torch.complex64, device=blocks.device)
            _check(L.gfdm_hip_cyclic_prefixer_add_device(self._h, self._dp(out, nb * F, "out"), self._dp(blocks, size, "in"), s, nb, self._sp(stream)))
            return out
        a = _c64(blocks)
        out = np.empty((nb, F), np.complex64)
        _check(L.gfdm_hip_cyclic_prefixer_add_host(self._h, out.ctypes.data, a.ctypes.data, s, nb))
        return out[0] if (np.asarray(blocks).ndim == 1 and nb == 1) else out

    generic_work = add_cyclic_prefix

    def remove_cyclic_prefix(self, frames, stream=None, out=None):
        L = lib()
        N, F = self.block_size(), self.frame_size()
        size, nb = self._batch(frames, F)
        if _is_tensor(frames):
            import torch
            if out is None:
                out = torch.empty(nb, N, dtype=torch.complex64, device=frames.device)
            _check(L.gfdm_hip_cyclic_prefixer_remove_device(self._h, self._dp(out, nb * N, "out"), self._dp(frames, size, "in"), nb, self._sp(stream)))
            return out
        a = _c64(frames)
        out = np.empty((nb, N), np.complex64)
        _check(L.gfdm_hip_cyclic_prefixer_remove_host(self._h, out.ctypes.data, a.ctypes.data, nb))
        return out[0] if (np.asarray(frames).ndim == 1 and nb == 1) else out


class ChannelEstimator(_Kernel):
    """preamble_channel_estimator_cc (include/gfdm/preamble_channel_estimator_cc.h:45-78; pybind name
    Preamble_channel_estimator, python/bindings/preamble_channel_estimator_python.cc): received preamble -> frame channel
    estimate.  Every call takes one preamble / estimate or a batch of them ([nframes][len]); numpy in -> numpy out (host path),
    torch in -> torch out (device path, current stream unless given)."""
    _destroy = "gfdm_hip_channel_estimator_destroy"

    def __init__(self, timeslots, fft_len, active_subcarriers, is_dc_free, which_estimator, preamble, device=0):
        L = lib()
        p = _c64(preamble).ravel()
        h = ctypes.c_void_p()
        _check(L.gfdm_hip_channel_estimator_create(ctypes.byref(h), timeslots, fft_len, active_subcarriers, int(bool(is_dc_free)),
                                                   int(which_estimator), p.ctypes.data, p.size, device))
        self._h = h
        self._dev = int(device)

    def timeslots(self):
        return lib().gfdm_hip_channel_estimator_timeslots(self._h)

    def fft_len(self):
        return lib().gfdm_hip_channel_estimator_fft_len(self._h)

    subcarriers = fft_len                   # the pybind property is called `subcarriers` (preamble_channel_estimator_python.cc:52)

    def active_subcarriers(self):
        return lib().gfdm_hip_channel_estimator_active_subcarriers(self._h)

    def frame_len(self):
        return lib().gfdm_hip_channel_estimator_frame_len(self._h)

    def is_dc_free(self):
        return bool(lib().gfdm_hip_channel_estimator_is_dc_free(self._h))

    def filtered_len(self):
        return lib().gfdm_hip_channel_estimator_filtered_len(self._h)

    def kernel_name(self):
        return lib().gfdm_hip_channel_estimator_kernel_name(self._h).decode()

    def preamble_filter_taps(self):
        out = np.empty(9, np.float32)
        lib().gfdm_hip_channel_estimator_preamble_filter_taps(self._h, out.ctypes.data)
        return out

    def _lens(self):
        return {"rx": 2 * self.fft_len(), "est": self.fft_len(), "filt": self.filtered_len(), "frame": self.frame_len()}

    def _run(self, name, x, n_in, n_out, stream):
        L = lib()
        if _is_tensor(x):
            import torch
            if x.numel() % n_in:
                raise RuntimeError("Input size %d is not a multiple of %d" % (x.numel(), n_in))
            nf = x.numel() // n_in
            out = torch.empty((nf, n_out) if x.dim() > 1 or nf != 1 else (n_out,), dtype=torch.complex64, device=x.device)
            _check(getattr(L, name + "_device")(self._h, out.data_ptr(), self._dp(x, nf * n_in, "in"), nf, self._sp(stream)))
            return out
        a = _c64(x)
        if a.size % n_in:
            raise RuntimeError("Input size %d is not a multiple of %d" % (a.size, n_in))
        nf = a.size // n_in
        out = np.empty((nf, n_out) if a.ndim > 1 or nf != 1 else (n_out,), np.complex64)
        _check(getattr(L, name + "_host")(self._h, out.ctypes.data, a.ctypes.data, nf))
        return out

    def estimate_frame(self, rx_preamble, stream=None):
        n = self._lens()
        return self._run("gfdm_hip_channel_estimator_estimate_frame", rx_preamble, n["rx"], n["frame"], stream)

    def estimate_preamble_channel(self, rx_preamble, stream=None):
        n = self._lens()
        return self._run("gfdm_hip_channel_estimator_estimate_preamble_channel", rx_preamble, n["rx"], n["est"], stream)

    def filter_preamble_estimate(self, estimate, stream=None):
        n = self._lens()
        return self._run("gfdm_hip_channel_estimator_filter_preamble_estimate", estimate, n["est"], n["filt"], stream)

    def interpolate_frame(self, filtered, stream=None):
        n = self._lens()
        return self._run("gfdm_hip_channel_estimator_interpolate_frame", filtered, n["filt"], n["frame"], stream)

    def prepare_for_zf(self, frame_estimate, stream=None):
        n = self._lens()
        return self._run("gfdm_hip_channel_estimator_prepare_for_zf", frame_estimate, n["frame"], n["frame"], stream)

    def estimate_snr(self, rx_preamble, stream=None):
        """(snr_lin, cnrs): scalars / [active] for one preamble, [nframes] / [nframes][active] for a batch."""
        L = lib()
        n_in, A = 2 * self.fft_len(), self.active_subcarriers()
        if _is_tensor(rx_preamble):
            import torch
            nf = rx_preamble.numel() // n_in
            snr = torch.empty(nf, dtype=torch.float32, device=rx_preamble.device)
            cnrs = torch.empty(nf, A, dtype=torch.float32, device=rx_preamble.device)
            _check(L.gfdm_hip_channel_estimator_estimate_snr_device(self._h, snr.data_ptr(), cnrs.data_ptr(), self._dp(rx_preamble, nf * n_in, "in"),
                                                                    nf, self._sp(stream)))
            return snr, cnrs
        a = _c64(rx_preamble)
        if a.size % n_in:
            raise RuntimeError("Input size %d is not a multiple of %d" % (a.size, n_in))
        nf = a.size // n_in
        snr = np.empty(nf, np.float32)
        cnrs = np.empty((nf, A), np.float32)
        _check(L.gfdm_hip_channel_estimator_estimate_snr_host(self._h, snr.ctypes.data, cnrs.ctypes.data, a.ctypes.data, nf))
        if a.ndim <= 1 and nf == 1:
            return float(snr[0]), cnrs[0]
        return snr, cnrs
