"""gfdm_amd: MI355X-native GFDM modulator / receiver / IC-receiver kernels (host-side Python mirror).

`Modulator`, `Demodulator`, `AdvancedReceiver` forward to the C-ABI of include/gfdm_hip.h
(libgfdm_hip.so).  `filters` generates prototype-filter taps.  Nothing here computes on the CPU.
"""
from . import filters  # noqa: F401
from .capi import (AdvancedReceiver, ChannelEstimator, CyclicPrefixer, Demodulator, GfdmHipError, Modulator, ResourceMapper, Transmitter, exported_symbols,  # noqa: F401
                   JIT_AUTO, JIT_BACKGROUND, JIT_IN_CONSTRUCTOR, JIT_OFF, generic_family_for_testing, lib, precompile, quiesce, set_dft_matrix_cores, set_ic_matrix_cores, set_jit,
                   HOST_COPY_ENGINES, HOST_ZERO_COPY, aligned_copy, aligned_empty, build_id, get_host_pipeline, host_call_stats, register_host, registered_host, set_host_pipeline, unregister_host)
