/* TEST DOUBLE -- not VOLK.  volk::vector<T> (an aligned std::vector) as lib/transmitter_cc_impl.h declares its scratch member. */
#ifndef MOCK_VOLK_ALLOC_HH
#define MOCK_VOLK_ALLOC_HH
#include <vector>
namespace volk {
template <class T> using vector = std::vector<T>;
} // namespace volk
#endif
