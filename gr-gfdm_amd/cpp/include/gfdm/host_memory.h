/* Registration of long-lived host buffers with the GPU kernels' host path (include/gfdm_hip.h, "the host-buffer batch path").
 *
 * The reference has no counterpart: its kernels run on the CPU and take any pointer.  Here a scheduler's buffers -- GNU Radio allocates a
 * block's circular buffers once, when the flowgraph starts -- can be pinned and mapped for the GPUs once; every later generic_work*(out, in)
 * on any part of them then runs IN PLACE on them (one launch across the PCIe link) instead of being bounced through pinned staging.
 * Buffers that are not registered keep working exactly as before, with the same results.
 *
 *     gr::gfdm::host_registration in_reg(in_base, in_bytes), out_reg(out_base, out_bytes);      // e.g. members of a block, made in start()
 *     ... work(): batched::sync_work(*d_kernel, noutput_items, in, out);                          // unchanged
 */
#ifndef INCLUDED_GFDM_HOST_MEMORY_H
#define INCLUDED_GFDM_HOST_MEMORY_H

#include <gfdm/api.h>
#include <cstddef>

namespace gr {
namespace gfdm {

class GFDM_API host_registration
{
public:
    /* pins [ptr, ptr + bytes) -- whole pages the caller owns: start and size multiples of the page size, as mmap'ed scheduler buffers are -- and maps
     * it for every GPU (~16 us per MiB); throws std::runtime_error when the range is not whole pages or the driver cannot pin it (the memory then
     * simply stays on the bounce path -- catching the exception is a valid way to run) */
    host_registration(void* ptr, std::size_t bytes);
    ~host_registration();                                   /* unregisters; the memory itself is the caller's */
    host_registration(const host_registration&) = delete;
    host_registration& operator=(const host_registration&) = delete;
    host_registration(host_registration&& other) noexcept : d_ptr(other.d_ptr) { other.d_ptr = nullptr; }
    void* data() const { return d_ptr; }

private:
    void* d_ptr;
};

} /* namespace gfdm */
} /* namespace gr */
#endif
