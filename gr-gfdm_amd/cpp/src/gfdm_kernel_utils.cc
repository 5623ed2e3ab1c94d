// gfdm_kernel_utils: shared helpers of the GPU-backed kernel classes.
#include <gfdm/gfdm_kernel_utils.h>
#include <gfdm_hip.h>

namespace gr {
namespace gfdm {

float gfdm_kernel_utils::calculate_signal_energy(const gfdm_complex* p_in, const int ninput_size)
{
    double acc = 0.0;
    for (int i = 0; i < ninput_size; ++i) acc += std::norm(p_in[i]);
    return static_cast<float>(acc);
}

namespace {
thread_local int t_default_device = 0;
}

int gfdm_kernel_utils::set_default_device(int device)
{
    const int prev = t_default_device;
    t_default_device = device;
    return prev;
}

int gfdm_kernel_utils::default_device() { return t_default_device; }

void gfdm_kernel_utils::throw_on_error(int status, const char* where)
{
    if (status == GFDM_HIP_OK) return;
    const char* detail = gfdm_hip_last_error();
    std::string msg = (detail && *detail) ? detail : gfdm_hip_strerror(status);
    if (status == GFDM_HIP_EINVAL_TAPS || status == GFDM_HIP_EINVAL_OVERLAP || status == GFDM_HIP_EINVAL)
        throw std::invalid_argument(msg);            // what the reference constructors throw
    throw std::runtime_error(std::string(where) + ": " + msg);
}

} // namespace gfdm
} // namespace gr
