/* TEST DOUBLE -- not GNU Radio; see io_signature.h in this directory. */
#ifndef MOCK_GNURADIO_SYNC_BLOCK_H
#define MOCK_GNURADIO_SYNC_BLOCK_H
#include <gnuradio/block.h>

namespace gr {
class sync_block : public block
{
public:
    virtual int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) = 0;

protected:
    sync_block();
    sync_block(const std::string& name, io_signature::sptr input_signature, io_signature::sptr output_signature);
};
} // namespace gr
#endif
