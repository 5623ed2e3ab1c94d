// GNU Radio block wrappers (gfdm/gr_blocks.h): a real translation unit only where GNU Radio's headers are installed; otherwise
// the header is empty and so is this file (no stand-in headers are used to make it build).
#include <gfdm/gr_blocks.h>

#ifdef GFDM_HAVE_GNURADIO
namespace gr {
namespace gfdm {
// anchor: instantiate the batched work() bodies for the three wrapper kernels
template int batched::sync_work<modulator_kernel_cc>(modulator_kernel_cc&, int, const batched::cfloat*, batched::cfloat*);
template int batched::sync_work<receiver_kernel_cc>(receiver_kernel_cc&, int, const batched::cfloat*, batched::cfloat*);
template int batched::sync_work_equalize<advanced_receiver_kernel_cc>(advanced_receiver_kernel_cc&, int, const batched::cfloat*,
                                                                     const batched::cfloat*, batched::cfloat*);
} // namespace gfdm
} // namespace gr
#endif
