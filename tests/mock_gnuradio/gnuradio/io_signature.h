/* TEST DOUBLE -- not GNU Radio.  Declares the handful of names of GNU Radio's public block API that gr-gfdm_amd's gfdm/gr_blocks.h
 * uses, so that tests/test_boundary.py can syntax-check that header (which is otherwise compiled only where GNU Radio is installed).
 * Written from the documented public API (gr::io_signature::make, gr::sync_block, gr::tag_t ...); nothing here computes anything and
 * nothing links against it.  Not used to build any reference code. */
#ifndef MOCK_GNURADIO_IO_SIGNATURE_H
#define MOCK_GNURADIO_IO_SIGNATURE_H
#include <complex>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

typedef std::complex<float> gr_complex;
typedef std::vector<const void*> gr_vector_const_void_star;
typedef std::vector<void*> gr_vector_void_star;

namespace gr {
class io_signature
{
public:
    typedef std::shared_ptr<io_signature> sptr;
    static sptr make(int min_streams, int max_streams, int sizeof_stream_item);
};
} // namespace gr
#endif
