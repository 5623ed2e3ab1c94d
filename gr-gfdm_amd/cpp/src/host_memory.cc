#include <gfdm/host_memory.h>
#include <gfdm_hip.h>

#include <stdexcept>
#include <string>

namespace gr {
namespace gfdm {

host_registration::host_registration(void* ptr, std::size_t bytes) : d_ptr(ptr)
{
    if (gfdm_hip_register_host(ptr, bytes) != GFDM_HIP_OK) {
        d_ptr = nullptr;
        throw std::runtime_error(std::string("host_registration: ") + gfdm_hip_last_error());
    }
}

host_registration::~host_registration()
{
    if (d_ptr) (void)gfdm_hip_unregister_host(d_ptr);
}

} // namespace gfdm
} // namespace gr
