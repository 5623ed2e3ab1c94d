// TEST-ONLY: what the loop-back "kernels" compute (loopback_kernels.cc) -- cheap, per element, and dependent on block index, operand, mode and kernel family, so that
// the driver (host_fuzz.cc) can tell a chunk that landed at the wrong offset, a stale staging set, a swapped operand or a family change inside one call from a
// correct result.  One definition for both sides: the comparison is bit for bit.
#ifndef GFDM_TEST_LOOPBACK_TRANSFORM_H
#define GFDM_TEST_LOOPBACK_TRANSFORM_H

namespace loopback {

// which kernel family served a launch (added to the real part of every output)
constexpr float kTagRowlane = 0.f, kTagJit = 0.25f, kTagGeneric = 0.5f;

struct c2 { float x, y; };

// receivers: sample s of the block, equaliser / preamble value e (0 when absent), mode 0 FD / 1 demod / 2 IC, ic_iter rounds
// (an advanced receiver with zero rounds IS the plain demodulator: the product launches that kernel for it, gfdm_jit.hip jit_launch_receive)
inline c2 rx_value(c2 s, c2 e, int mode, int ic_iter, float tag)
{
    const int m = (mode == 2 && ic_iter == 0) ? 1 : mode;
    return c2{ s.x * (float)(1 + m) + e.x + tag, s.y - e.y + (float)ic_iter };
}
// modulator, bare blocks
inline c2 mod_value(c2 s, float tag) { return c2{ s.x * 3.f + tag, s.y + 1.f }; }
// transmitter: sample j of the frame of port `port`, from symbol s
inline c2 tx_value(c2 s, int port, int framed, float tag) { return c2{ s.x + (float)port + tag, s.y + (framed ? 2.f : 4.f) }; }
inline c2 td_value(c2 s) { return c2{ s.x * 5.f, s.y }; }
inline c2 cancel_value(c2 td, c2 fd) { return c2{ td.x * 7.f + fd.x, td.y - fd.y }; }
inline c2 est_value(c2 s, int in_stage, int out_stage, float tag) { return c2{ s.x * 2.f + (float)(4 * in_stage + out_stage) + tag, s.y }; }

}  // namespace loopback

#endif
