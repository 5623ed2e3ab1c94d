/* Minimal constellation object for the IC receiver.
 *
 * gr-gfdm's advanced_receiver_kernel_cc takes a gr::digital::constellation_sptr (GNU Radio's
 * gr-digital, include/gfdm/advanced_receiver_kernel_cc.h:25,46) and uses exactly two members:
 * points() and decision_maker() (lib/advanced_receiver_kernel_cc.cc:114,119).  GNU Radio is not a
 * dependency of this library, so the same information travels as a points array plus a decision
 * rule.  When GNU Radio headers are available, advanced_receiver_kernel_cc.h also offers the
 * reference constructor and converts the sptr with from_points().
 */
#ifndef INCLUDED_GFDM_CONSTELLATION_H
#define INCLUDED_GFDM_CONSTELLATION_H

#include <gfdm/api.h>
#include <complex>
#include <memory>
#include <vector>

namespace gr {
namespace gfdm {

class GFDM_API constellation
{
public:
    enum decision_rule { AUTO = -1, NEAREST = 0, QPSK = 1, BPSK = 2 };

    constellation(std::vector<std::complex<float>> points, decision_rule rule = AUTO) : d_points(std::move(points)), d_rule(rule) {}

    /* gr::digital::constellation_qpsk: (-1-1j, 1-1j, -1+1j, 1+1j)/sqrt(2), index 2*(im>0)+(re>0) */
    static std::shared_ptr<constellation> qpsk();
    /* gr::digital::constellation_bpsk: (-1, 1), index (re>0) */
    static std::shared_ptr<constellation> bpsk();
    static std::shared_ptr<constellation> from_points(std::vector<std::complex<float>> points) { return std::make_shared<constellation>(std::move(points), AUTO); }

    const std::vector<std::complex<float>>& points() const { return d_points; }
    decision_rule rule() const { return d_rule; }
    /* host-side equivalent of decision_maker(); the GPU kernels implement the same rule */
    unsigned int decision_maker(const std::complex<float>* sample) const;

private:
    std::vector<std::complex<float>> d_points;
    decision_rule d_rule;
};

typedef std::shared_ptr<constellation> constellation_sptr;

} // namespace gfdm
} // namespace gr

#endif /* INCLUDED_GFDM_CONSTELLATION_H */
