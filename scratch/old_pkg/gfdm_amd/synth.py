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


def channel_response(block_start, nblocks, block_size, device):
    """Per-block one-tap equaliser input f_eq[b] = FFT_N(h) * exp(0.01j * b), complex64 (nblocks, block_size)."""
    h = torch.zeros(block_size, dtype=torch.complex64, device=device)
    h[:len(CHANNEL)] = torch.tensor(CHANNEL, dtype=torch.complex64, device=device)
    H = torch.fft.fft(h)
    b = torch.arange(block_start, block_start + nblocks, dtype=torch.float32, device=device)
    phase = torch.polar(torch.ones_like(b), 0.01 * b)
    return (phase[:, None] * H[None, :]).contiguous()


def through_channel(frames, f_eq):
    """Apply the per-block circular channel to modulated frames (input preparation, not timed)."""
    return torch.fft.ifft(torch.fft.fft(frames, dim=-1) * f_eq, dim=-1).to(torch.complex64).contiguous()
