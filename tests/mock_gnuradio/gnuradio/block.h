/* TEST DOUBLE -- not GNU Radio; see io_signature.h in this directory.  gr::block as far as gfdm/gr_blocks.h and the syntax check of the
 * reference's unchanged *_impl.cc wrappers (tests/test_boundary.py) use it: declarations only. */
#ifndef MOCK_GNURADIO_BLOCK_H
#define MOCK_GNURADIO_BLOCK_H
#include <gnuradio/io_signature.h>
#include <pmt/pmt.h>
#include <cstring>
#include <cassert>      /* GNU Radio's own headers bring it in; lib/transmitter_cc_impl.cc:159 relies on that */

typedef std::vector<int> gr_vector_int;

namespace gr {
struct tag_t {
    uint64_t offset;
    pmt::pmt_t key, value;
};

class block
{
public:
    enum tag_propagation_policy_t { TPP_DONT = 0, TPP_ALL_TO_ALL = 1, TPP_ONE_TO_ONE = 2 };
    virtual ~block();
    void set_output_multiple(int multiple);
    void set_tag_propagation_policy(tag_propagation_policy_t p);
    void add_item_tag(unsigned int which_output, const tag_t& tag);
    void add_item_tag(unsigned int which_output, uint64_t abs_offset, const pmt::pmt_t& key, const pmt::pmt_t& value);
    void remove_item_tag(unsigned int which_input, const tag_t& tag);
    void get_tags_in_window(std::vector<tag_t>& v, unsigned int which_input, uint64_t rel_start, uint64_t rel_end);
    void get_tags_in_window(std::vector<tag_t>& v, unsigned int which_input, uint64_t rel_start, uint64_t rel_end, const pmt::pmt_t& key);
    void get_tags_in_range(std::vector<tag_t>& v, unsigned int which_input, uint64_t abs_start, uint64_t abs_end, const pmt::pmt_t& key);
    uint64_t nitems_read(unsigned int which_input);
    uint64_t nitems_written(unsigned int which_output);
    void consume_each(int how_many_items);
    void set_relative_rate(double relative_rate);
    void set_fixed_rate(bool fixed_rate);
    virtual void forecast(int noutput_items, gr_vector_int& ninput_items_required);
    virtual int fixed_rate_ninput_to_noutput(int ninput);
    virtual int fixed_rate_noutput_to_ninput(int noutput);
    virtual int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                             gr_vector_void_star& output_items);

protected:
    block();      // the reference's interface classes inherit virtually (class transmitter_cc : virtual public gr::block)
    block(const std::string& name, io_signature::sptr input_signature, io_signature::sptr output_signature);
};
} // namespace gr

/* gnuradio::make_block_sptr<T>(args...) (GNU Radio >= 3.9, gnuradio/sptr_magic.h): what every ::make() of the reference calls */
namespace gnuradio {
template <class T, class... Args>
std::shared_ptr<T> make_block_sptr(Args&&... args);
} // namespace gnuradio
#endif
