"""gfdm_amd: MI355X-native GFDM modulator / receiver / IC-receiver kernels (host-side Python mirror).

`Modulator`, `Demodulator`, `AdvancedReceiver` forward to the C-ABI of include/gfdm_hip.h
(libgfdm_hip.so).  `filters` generates prototype-filter taps.  Nothing here computes on the CPU.
"""
from . import filters  # noqa: F401
from .capi import (AdvancedReceiver, ChannelEstimator, CyclicPrefixer, Demodulator, GfdmHipError, Modulator, ResourceMapper, Transmitter, exported_symbols,  # noqa: F401
                   generic_family_for_testing, lib, set_ic_matrix_cores, set_jit)
