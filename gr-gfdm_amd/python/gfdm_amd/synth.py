"""Synthetic GFDM workloads generated in place on the device.

QPSK symbols come from a counter-based generator keyed on (seed, global block index, symbol index), so a batch
sharded over several GPUs is bit-identical to the same batch generated on one.  torch is used as plumbing only
(device memory + elementwise integer ops); nothing here is on the timed path.
"""
import numpy as np
import torch

SEED = 0x6FD1
CHANNEL = (1.0, 0.5, 0.1j, 0.1 + 0.05j)          # 4-tap test channel of python/qa_python_bindings.py:468

_M1 = -7046029254386353131                       # 0x9E3779B97F4A7C15 as int64
_M2 = -4658895280553007687                       # 0xBF58476D1CE4E5B9
_M3 = -7723592293110705685                       # 0x94D049BB133111EB


def _mix64(x):
    """splitmix64 finaliser on int64 tensors (wrap-around arithmetic; logical shifts emulated by masking)."""
    def lsr(v, s):
        return (v >> s) & ((1 << (64 - s)) - 1)
    x = (x ^ lsr(x, 30)) * _M2
    x = (x ^ lsr(x, 27)) * _M3
    return x ^ lsr(x, 31)


def qpsk_symbols(block_start, nblocks, block_size, device, seed=SEED, active_mask=None):
    """(nblocks, block_size) complex64 QPSK symbols (+-1 +-1j)/sqrt(2) for global blocks [block_start, block_start+nblocks)."""
    b = torch.arange(block_start, block_start + nblocks, dtype=torch.int64, device=device)[:, None]
    i = torch.arange(block_size, dtype=torch.int64, device=device)[None, :]
    h = _mix64((b * block_size + i) * _M1 + seed)
    s = np.float32(np.sqrt(0.5))
    re = torch.where((h & 1) != 0, -s, s).to(torch.float32)
    im = torch.where((h & 2) != 0, -s, s).to(torch.float32)
    out = torch.complex(re, im)
    if active_mask is not None:
        out = out * active_mask.to(out.dtype)
    return out.contiguous()


def _block_phase(block_start, nblocks):
    """exp(0.01j b) for global blocks b, computed on the host in float64 (numpy): (cos, sin) as float32 arrays"""
    b = np.arange(block_start, block_start + nblocks, dtype=np.float64)
    return np.cos(0.01 * b).astype(np.float32), np.sin(0.01 * b).astype(np.float32)


def _cmul_parts(ar, ai, br, bi):
    """(ar + j ai)(br + j bi) from single-rounding real operations (every torch op below is ONE IEEE operation per element, so the
    result does not depend on which code path -- vectorised, tail, fused-multiply-add contracted or not -- an element-wise kernel takes
    for a given tensor size: a batch generated in one piece equals the same blocks generated shard by shard, bit for bit)"""
    return ar * br - ai * bi, ar * bi + ai * br


def channel_response(block_start, nblocks, block_size, device):
    """Per-block one-tap equaliser input f_eq[b] = FFT_N(h) * exp(0.01j * b), complex64 (nblocks, block_size); block b's vector is the
    same whatever batch it is generated in."""
    h = np.zeros(block_size, dtype=np.complex128)
    h[:len(CHANNEL)] = CHANNEL
    H = np.fft.fft(h).astype(np.complex64)
    c, s = _block_phase(block_start, nblocks)
    Hr = torch.from_numpy(np.ascontiguousarray(H.real)).to(device)[None, :]
    Hi = torch.from_numpy(np.ascontiguousarray(H.imag)).to(device)[None, :]
    re, im = _cmul_parts(torch.from_numpy(c).to(device)[:, None], torch.from_numpy(s).to(device)[:, None], Hr, Hi)
    return torch.complex(re, im).contiguous()


def through_channel(frames, f_eq):
    """Apply the per-block circular channel to modulated frames (input preparation, not timed)."""
    return torch.fft.ifft(torch.fft.fft(frames, dim=-1) * f_eq, dim=-1).to(torch.complex64).contiguous()


def through_test_channel(frames, block_start):
    """The frames of global blocks [block_start, ...) through the circular test channel whose response channel_response() returns -- in the
    TIME domain (y[n] = exp(0.01j b) * sum_t h[t] x[(n - t) mod N], four rolled multiply-adds out of single-rounding real operations), so
    that block b's result does not depend on which batch it is generated in (a batched FFT picks its plan by the batch size, and an
    element-wise complex multiply its code path by the tensor size; the strong-scaled bench configurations compare output checksums
    between different splits of the same global blocks)."""
    nblocks = frames.shape[0]
    c, s = _block_phase(block_start, nblocks)
    xr, xi = frames.real, frames.imag
    out_r = torch.zeros_like(xr)
    out_i = torch.zeros_like(xi)
    for t, h in enumerate(CHANNEL):
        coef = (c.astype(np.float64) + 1j * s.astype(np.float64)) * complex(h)         # per-block coefficient, rounded once to float32
        cr = torch.from_numpy(coef.real.astype(np.float32)).to(frames.device)[:, None]
        ci = torch.from_numpy(coef.imag.astype(np.float32)).to(frames.device)[:, None]
        tr, ti = _cmul_parts(cr, ci, torch.roll(xr, t, dims=-1), torch.roll(xi, t, dims=-1))
        out_r = out_r + tr
        out_i = out_i + ti
    return torch.complex(out_r, out_i).contiguous()
