"""BASELINE configs[3] and configs[4] at the batch they STATE -- 65 536 blocks -- in one launch on one GPU.

configs[4] (K=256, M=31) is 63 488 bytes per block: one buffer is 4.16 GB and the byte offset of a block passes 2**31 inside block
33 825 (33 825 * 63 488 = 2 147 481 600), so blocks 33 825 / 33 826 are the first whose addresses no 32-bit offset reaches;
configs[3] (K=128, M=15) stays under 2**31 (1.0 GB) and is here for its stated batch.  The float64 oracle cannot run 65 536 blocks, so:

  * sampled blocks -- 0, 33 825, 33 826, 65 535 and twelve random ones -- against oracle/gfdm_ref.py for modulate, MF, ZF, MF + 2 IC and
    ZF + 2 IC (semantics: lib/modulator_kernel_cc.cc:98-141, lib/receiver_kernel_cc.cc:301-334, lib/advanced_receiver_kernel_cc.cc:93-107),
    1e-5 relative L2 per block (BASELINE.json north_star);
  * over the WHOLE batch: the one 65 536-block launch equals eight 8192-block launches of the same handle bit for bit (blocks are
    independent; the 8192-block launches are what tests/test_parity_gpu.py::test_full_size_properties checks), the IC receivers return every
    transmitted QPSK symbol's quadrant, receiver linearity, equaliser round trip.
Inputs are generated on the device (gfdm_amd.synth, counter-based: block b is the same whatever chunk generates it).
"""
import numpy as np
import pytest

import gfdm_ref as R
from conftest import check_err, have_gpu, rel_err
from gfdm_amd.filters import get_frequency_domain_filter
from test_parity_gpu import guarded

pytestmark = pytest.mark.gpu

TOL = 1e-5
CHUNK = 8192
STATED = [("cfg4", 15, 128, 4, 0.2, 65536), ("cfg5", 31, 256, 2, 0.1, 65536)]


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    if not have_gpu():
        pytest.fail("no MI355X visible: the HIP path cannot run (there is no CPU fallback to test instead)")


@pytest.mark.parametrize("name,M,K,L,alpha,B", STATED)
def test_stated_batch_in_one_launch(name, M, K, L, alpha, B):
    import torch
    import gfdm_amd
    from gfdm_amd import synth
    dev = torch.device("cuda:0")
    N = M * K
    taps = get_frequency_domain_filter("rrc", alpha, M, K, L)
    nt = R.normalize_taps(taps, M)
    allk = np.arange(K)
    mod, dem = gfdm_amd.Modulator(M, K, L, taps), gfdm_amd.Demodulator(M, K, L, taps)
    adv = gfdm_amd.AdvancedReceiver(M, K, L, taps, allk, 2, R.qpsk_points())
    assert (mod.kernel_name(), dem.kernel_name(), adv.kernel_name()) == ("rowlane",) * 3
    new = lambda: torch.empty(B, N, dtype=torch.complex64, device=dev)

    sym, feq = new(), new()
    for c in range(0, B, CHUNK):
        sym[c:c + CHUNK] = synth.qpsk_symbols(c, CHUNK, N, dev)
        feq[c:c + CHUNK] = synth.channel_response(c, CHUNK, N, dev)
    # ---- every mode, ONE launch over the 65 536 blocks
    x = mod.modulate(sym, out=new())
    xe = new()
    for c in range(0, B, CHUNK):                           # (input preparation through torch.fft: chunked to bound its temporaries)
        xe[c:c + CHUNK] = synth.through_channel(x[c:c + CHUNK], feq[c:c + CHUNK])
    y = dem.demodulate(x, out=new())
    ye = dem.demodulate_equalize(xe, feq, out=new())
    zmf = adv.demodulate(x, out=new())
    zzf = adv.demodulate_equalize(xe, feq, out=new())
    torch.cuda.synchronize()

    # ---- sampled blocks against the float64 oracle
    edge = [0, 33825, 33826, B - 1]                        # 33 825 straddles byte offset 2**31 of configs[4]'s buffers, 33 826 is the first block past it
    pick = np.unique(np.concatenate((edge, np.random.default_rng(65536 + K).integers(0, B, 12))))
    idx = torch.as_tensor(pick, device=dev)
    h = lambda t: t[idx].cpu().numpy()
    sym_h, x_h, xe_h, feq_h = h(sym), h(x), h(xe), h(feq)
    check_err("%s_65536_modulate" % name, rel_err(x_h, R.modulate(sym_h, nt, M, K, L)), TOL)
    check_err("%s_65536_demod_mf" % name, rel_err(h(y), R.demodulate(x_h, nt, M, K, L)), TOL)
    check_err("%s_65536_demod_zf" % name, rel_err(h(ye), R.demodulate(xe_h, nt, M, K, L, f_eq=feq_h)), TOL)
    for tag, got, inp, eq in (("mf_ic2", zmf, x_h, None), ("zf_ic2", zzf, xe_h, feq_h)):
        ref, st = R.advanced_receive(inp, nt, M, K, L, allk, R.qpsk_points(), 2, f_eq=eq, kind="qpsk", return_stages=True)
        keep = guarded(st, allk, K, M)
        assert keep.sum() >= len(pick) - 1 and keep[np.searchsorted(pick, edge)].all()
        check_err("%s_65536_%s" % (name, tag), rel_err(h(got)[keep], ref[keep]), TOL)

    # ---- whole batch: one launch == eight 8192-block launches, bit for bit
    part = torch.empty(CHUNK, N, dtype=torch.complex64, device=dev)
    runs = ((x, lambda c: mod.modulate(sym[c:c + CHUNK], out=part)),
            (y, lambda c: dem.demodulate(x[c:c + CHUNK], out=part)),
            (ye, lambda c: dem.demodulate_equalize(xe[c:c + CHUNK], feq[c:c + CHUNK], out=part)),
            (zmf, lambda c: adv.demodulate(x[c:c + CHUNK], out=part)),
            (zzf, lambda c: adv.demodulate_equalize(xe[c:c + CHUNK], feq[c:c + CHUNK], out=part)))
    for whole, run in runs:
        for c in range(0, B, CHUNK):
            run(c)
            assert torch.equal(whole[c:c + CHUNK].view(torch.float32), part.view(torch.float32)), c
    del part

    # ---- whole batch: IC loop-back, every symbol of every block in the transmitted quadrant and close to the point
    for z in (zmf, zzf):
        assert bool(torch.all(torch.signbit(z.real) == torch.signbit(sym.real))) and bool(torch.all(torch.signbit(z.imag) == torch.signbit(sym.imag)))
    assert float((zzf - sym).abs().max()) < 0.2
    del zmf, zzf
    # equaliser round trip (fp32 channel application + division: 1e-4 is its own noise floor)
    ymax = float(y.abs().max())
    assert float((ye - y).abs().max()) / ymax < 1e-4
    del ye, xe, feq
    # receiver linearity across the two halves of the batch (blocks below and above the 2**31 boundary mixed with each other)
    a, b = 0.75 - 0.5j, -1.25 + 0.25j
    mix = a * x + b * x.flip(0)
    lin = dem.demodulate(mix, out=mix)                      # in place: the kernel reads a block before it writes it
    want = a * y + b * y.flip(0)
    assert float((lin - want).abs().max()) / ymax < 2e-5
