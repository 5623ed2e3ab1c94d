/* gr::gfdm::transmitter_kernel -- public interface of gr-gfdm's include/gfdm/transmitter_kernel.h:43-85
 * (resource mapper -> modulator -> cyclic prefix/suffix + ramp -> preamble), executed as ONE fused HIP kernel
 * behind include/gfdm_hip.h.  Drop-in for lib/transmitter_cc_impl.cc:61-70,165-177.
 */
#ifndef INCLUDED_GFDM_TRANSMITTER_KERNEL_H
#define INCLUDED_GFDM_TRANSMITTER_KERNEL_H

#include <gfdm/gfdm_kernel_utils.h>

struct gfdm_hip_transmitter;

namespace gr {
namespace gfdm {

class GFDM_API transmitter_kernel
{
public:
    typedef gr::gfdm::gfdm_kernel_utils::gfdm_complex gfdm_complex;

    /* throws std::invalid_argument for the argument errors of the reference's mapper / modulator / prefixer /
     * transmitter constructors (same messages), std::runtime_error when no GPU is usable */
    transmitter_kernel(int timeslots,
                       int subcarriers,
                       int active_subcarriers,
                       int cp_len,
                       int cs_len,
                       int ramp_len,
                       std::vector<int> subcarrier_map,
                       bool per_timeslot,
                       int overlap,
                       std::vector<gfdm_complex> frequency_taps,
                       std::vector<gfdm_complex> window_taps,
                       std::vector<int> cyclic_shifts,
                       std::vector<std::vector<gfdm_complex>> preambles);
    ~transmitter_kernel();
    transmitter_kernel(const transmitter_kernel&) = delete;
    transmitter_kernel& operator=(const transmitter_kernel&) = delete;

    int input_vector_size();
    int output_vector_size();
    /* one frame for cyclic_shifts()[0] */
    void generic_work(gfdm_complex* p_out, const gfdm_complex* p_in, const int ninput_size);
    /* mapper + modulator: one bare block */
    void modulate(gfdm_complex* out, const gfdm_complex* in, const int ninput_size);
    /* preamble of `cyclic_shift` + cyclic prefix/suffix + ramp around an already modulated block */
    void add_frame(gfdm_complex* out, const gfdm_complex* in, const int cyclic_shift);
    const std::vector<int>& cyclic_shifts() const { return d_cyclic_shifts; }

    /* --- additions: whole batches and all ports per call --- */
    /* nframes frames of ninput_size symbols each; outs[i] receives nframes frames for cyclic_shifts()[i], i < n_ports */
    void generic_work_batch(gfdm_complex* const* outs, int n_ports, const gfdm_complex* in, int ninput_size, long nframes);
    void generic_work_device(void* const* d_outs, int n_ports, const void* d_in, int ninput_size, long nframes, void* hip_stream);
    const char* kernel_name() const;

private:
    std::vector<int> d_cyclic_shifts;
    gfdm_hip_transmitter* d_handle;
};

} // namespace gfdm
} // namespace gr

#endif /* INCLUDED_GFDM_TRANSMITTER_KERNEL_H */
