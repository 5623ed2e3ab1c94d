/* TEST DOUBLE -- not GNU Radio's PMT library; see gnuradio/io_signature.h in this directory.  Declarations only. */
#ifndef MOCK_PMT_PMT_H
#define MOCK_PMT_PMT_H
#include <cstddef>
#include <memory>
#include <string>
#include <vector>

namespace pmt {
class pmt_base;
typedef std::shared_ptr<pmt_base> pmt_t;
pmt_t string_to_symbol(const std::string& s);
pmt_t intern(const std::string& s);
pmt_t from_long(long x);
pmt_t from_float(double x);
pmt_t init_f32vector(size_t k, const float* data);
pmt_t init_f32vector(size_t k, const std::vector<float>& data);
long to_long(pmt_t x);
bool eqv(const pmt_t& x, const pmt_t& y);
const std::string symbol_to_string(const pmt_t& sym);
} // namespace pmt
#endif
