"""Prototype-filter tap generation for GFDM (host-side helper).

Plays the role of the reference's python/pygfdm/filters.py:27-54
(`get_frequency_domain_filter`): root-raised-cosine / raised-cosine impulse
response of length M*K, rolled so the peak sits on sample 0, FFT'd, reduced
to the L*M bins around DC in the order [H[0:ML/2], H[-ML/2:]] and scaled to
energy M.  The reference takes the impulse response from scikit-commpy
(un-vendored, not installed here); this file evaluates the textbook closed
forms itself, so tap VALUES are not pinned against commpy.  Taps are an input
to the kernels: every parity test feeds identical taps to both sides.
"""
import numpy as np

__all__ = ["rrc_impulse", "rc_impulse", "gfdm_filter_taps", "gfdm_freq_taps",
           "gfdm_freq_taps_sparse", "get_frequency_domain_filter"]


def rrc_impulse(n, alpha, ts):
    """Root-raised-cosine impulse response, n samples centred on n/2, symbol period ts samples."""
    t = (np.arange(n, dtype=np.float64) - n / 2.0)
    h = np.empty(n, dtype=np.float64)
    x = t / ts
    with np.errstate(divide="ignore", invalid="ignore"):
        num = np.sin(np.pi * x * (1.0 - alpha)) + 4.0 * alpha * x * np.cos(np.pi * x * (1.0 + alpha))
        den = np.pi * x * (1.0 - (4.0 * alpha * x) ** 2)
        h = num / den
    zero = t == 0.0
    h[zero] = 1.0 - alpha + 4.0 * alpha / np.pi
    if alpha != 0.0:
        sing = np.isclose(np.abs(x), 1.0 / (4.0 * alpha), rtol=0.0, atol=1e-12)
        h[sing] = (alpha / np.sqrt(2.0)) * ((1.0 + 2.0 / np.pi) * np.sin(np.pi / (4.0 * alpha))
                                           + (1.0 - 2.0 / np.pi) * np.cos(np.pi / (4.0 * alpha)))
    return h


def rc_impulse(n, alpha, ts):
    """Raised-cosine impulse response, n samples centred on n/2."""
    t = (np.arange(n, dtype=np.float64) - n / 2.0)
    x = t / ts
    with np.errstate(divide="ignore", invalid="ignore"):
        h = np.sinc(x) * np.cos(np.pi * alpha * x) / (1.0 - (2.0 * alpha * x) ** 2)
    if alpha != 0.0:
        sing = np.isclose(np.abs(x), 1.0 / (2.0 * alpha), rtol=0.0, atol=1e-12)
        h[sing] = (np.pi / 4.0) * np.sinc(1.0 / (2.0 * alpha))
    return h


def gfdm_filter_taps(filtertype, alpha, M, K, oversampling_factor=1):
    n = int(M * K * oversampling_factor)
    ts = float(K * oversampling_factor)
    if filtertype == "rrc":
        return rrc_impulse(n, alpha, ts)
    if filtertype == "rc":
        return rc_impulse(n, alpha, ts)
    raise ValueError("filtertype must be 'rrc' or 'rc'")


def gfdm_freq_taps(h):
    return np.fft.fft(np.roll(h, h.shape[-1] // 2))


def gfdm_freq_taps_sparse(H, M, L):
    return np.concatenate((H[0:(M * L) // 2], H[-(M * L) // 2:]))


def get_frequency_domain_filter(filtertype, alpha, M, K, L):
    """Sparse frequency-domain taps, energy-M normalised, complex128, length L*M."""
    H = gfdm_freq_taps_sparse(gfdm_freq_taps(gfdm_filter_taps(filtertype, alpha, M, K, 1)), M, L)
    H = H / np.sqrt((H * np.conj(H)).sum().real / M)
    return H
