/* TEST DOUBLE -- not gr-digital; see ../io_signature.h.  The two members of gr::digital::constellation that gr-gfdm calls
 * (lib/advanced_receiver_kernel_cc.cc:114,119), declarations only. */
#ifndef MOCK_GNURADIO_DIGITAL_CONSTELLATION_H
#define MOCK_GNURADIO_DIGITAL_CONSTELLATION_H
#include <gnuradio/digital/api.h>
#include <gnuradio/io_signature.h>

namespace gr {
namespace digital {
class constellation
{
public:
    virtual ~constellation();
    std::vector<gr_complex> points();
    virtual unsigned int decision_maker(const gr_complex* sample) = 0;
};
typedef std::shared_ptr<constellation> constellation_sptr;
} // namespace digital
} // namespace gr
#endif
