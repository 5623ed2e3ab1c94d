// pybind11 module `gfdm_python`: the reference's kernel-level Python surface
// (python/bindings/python_bindings.cc:52, modulator_python.cc:34-59, demodulator_python.cc:35-205)
// on top of the GPU-backed classes, plus an AdvancedReceiver kernel binding and batched calls.
// Unlike the reference module it does not import gnuradio.gr.
#include <pybind11/complex.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <gfdm/add_cyclic_prefix_cc.h>
#include <gfdm/advanced_receiver_kernel_cc.h>
#include <gfdm/modulator_kernel_cc.h>
#include <gfdm/receiver_kernel_cc.h>
#include <gfdm/resource_mapper_kernel_cc.h>
#include <gfdm/preamble_channel_estimator_cc.h>
#include <gfdm/transmitter_kernel.h>
#include <gfdm_hip.h>


namespace py = pybind11;
using namespace gr::gfdm;

typedef std::complex<float> cfloat;
typedef py::array_t<cfloat, py::array::c_style | py::array::forcecast> carray;

namespace {

// The size/shape checks and their messages are part of the reference's behaviour
// (python/bindings/modulator_python.cc:44-52, demodulator_python.cc:49-57,118-132).
void require_1d(const py::buffer_info& a) { if (a.ndim != 1) throw std::runtime_error("Only ONE-dimensional vectors allowed!"); }

void require_size(const py::buffer_info& a, int block_size, const char* what, const char* owner)
{
    if (a.size != block_size)
        throw std::runtime_error(std::string(what) + " vector size(" + std::to_string(a.size) + ") MUST be equal to " + owner +
                                 ".block_size(" + std::to_string(block_size) + ")!");
}

void require_batch(const py::buffer_info& a, int block_size, const char* what)
{
    if (block_size <= 0 || a.size % block_size)
        throw std::runtime_error(std::string(what) + " size(" + std::to_string(a.size) + ") MUST be a multiple of block_size(" +
                                 std::to_string(block_size) + ")!");
}

py::array_t<cfloat> like(const py::buffer_info& a) { return py::array_t<cfloat>(a.shape); }

cfloat* ptr(py::buffer_info& b) { return static_cast<cfloat*>(b.ptr); }
const cfloat* cptr(const py::buffer_info& b) { return static_cast<const cfloat*>(b.ptr); }

// one-input single-block method of the demodulator (messages say "Modulator", as the reference does for these three)
template <typename Fn>
py::array_t<cfloat> rx_unary(receiver_kernel_cc& self, const carray& array, Fn fn)
{
    py::buffer_info in = array.request();
    require_1d(in);
    require_size(in, self.block_size(), "Input", "Modulator");
    auto result = py::array_t<cfloat>(in.size);
    py::buffer_info out = result.request();
    fn(ptr(out), cptr(in));
    return result;
}

template <typename Fn>
py::array_t<cfloat> rx_binary(receiver_kernel_cc& self, const carray& array, const carray& second, Fn fn)
{
    py::buffer_info in = array.request(), in2 = second.request();
    if (in.ndim != 1 || in2.ndim != 1) throw std::runtime_error("Only ONE-dimensional vectors allowed!");
    require_size(in, self.block_size(), "Input", "Demodulator");
    require_size(in2, self.block_size(), "Channel", "Demodulator");
    auto result = py::array_t<cfloat>(in.size);
    py::buffer_info out = result.request();
    fn(ptr(out), cptr(in), cptr(in2));
    return result;
}

// frames of frame_len samples in, (nframes, noutput) demapped symbols out.  Input stride and output size come from the library
// (io_layout): it alone knows what the kernels write for this handle's configuration and this noutput_size.
template <typename Kernel>
py::array_t<cfloat> run_frames(Kernel& self, const carray& frames, py::object eq, int noutput_size)
{
    const auto ly = self.io_layout(false, noutput_size);
    py::buffer_info in = frames.request();
    if (ly.n_in <= 0 || in.size % ly.n_in)
        throw std::runtime_error("frames size(" + std::to_string(in.size) + ") MUST be a multiple of frame_len(" + std::to_string(ly.n_in) + ")!");
    const long nframes = in.size / ly.n_in;
    py::array_t<cfloat> result(std::vector<py::ssize_t>{ nframes, ly.n_out });
    py::buffer_info out = result.request();
    if (eq.is_none()) {
        self.generic_work_frames_batch(ptr(out), cptr(in), nullptr, noutput_size, nframes);
    } else {
        carray eq_arr = eq.cast<carray>();
        py::buffer_info e = eq_arr.request();
        if (e.size != nframes * self.block_size())
            throw std::runtime_error("Channel vector size(" + std::to_string(e.size) + ") MUST be equal to nframes * block_size!");
        self.generic_work_frames_batch(ptr(out), cptr(in), cptr(e), noutput_size, nframes);
    }
    return result;
}

// blocks (or frames, when configure_frames was called) + received preambles in, symbols out: the estimator runs inside the kernel
template <typename Kernel>
py::array_t<cfloat> run_estimated(Kernel& self, const carray& x, const carray& pre, int preamble_stride, int noutput_size)
{
    const auto ly = self.io_layout(true, noutput_size);
    py::buffer_info in = x.request(), p = pre.request();
    if (ly.n_in <= 0 || in.size == 0 || in.size % ly.n_in)
        throw std::runtime_error("Input size(" + std::to_string(in.size) + ") MUST be a multiple of " + std::to_string(ly.n_in) + "!");
    const long nblocks = in.size / ly.n_in;
    const long stride = preamble_stride > 0 ? preamble_stride : 2 * ly.est_fft_len;
    if (p.size < (nblocks - 1) * stride + 2 * ly.est_fft_len)
        throw std::runtime_error("rx_preamble size(" + std::to_string(p.size) + ") MUST be at least " +
                                 std::to_string((nblocks - 1) * stride + 2 * ly.est_fft_len) + "!");
    py::array_t<cfloat> result(std::vector<py::ssize_t>{ nblocks, ly.n_out });
    py::buffer_info out = result.request();
    self.generic_work_estimated_batch(ptr(out), cptr(in), cptr(p), preamble_stride, noutput_size, nblocks);
    return result;
}

} // namespace

PYBIND11_MODULE(gfdm_python, m)
{
    m.doc() = "GFDM modulator / receiver kernels on AMD MI355X (HIP) behind gr-gfdm's kernel-class API";
    // background builds of run-time instantiated kernels must not outlive the interpreter's tear-down (gfdm_hip_quiesce, include/gfdm_hip.h)
    py::module_::import("atexit").attr("register")(py::cpp_function([]() { gfdm_hip_quiesce(); }));

    py::class_<modulator_kernel_cc>(m, "Modulator")
        .def(py::init<int, int, int, std::vector<cfloat>>(), py::arg("timeslots"), py::arg("subcarriers"), py::arg("overlap"),
             py::arg("frequency_taps"))
        .def("block_size", &modulator_kernel_cc::block_size)
        .def("filter_taps", &modulator_kernel_cc::filter_taps)
        .def("kernel_name", &modulator_kernel_cc::kernel_name)
        .def("modulate",
             [](modulator_kernel_cc& self, const carray array) {
                 py::buffer_info in = array.request();
                 require_1d(in);
                 require_size(in, self.block_size(), "Input", "Modulator");
                 auto result = py::array_t<cfloat>(in.size);
                 py::buffer_info out = result.request();
                 self.generic_work(ptr(out), cptr(in));
                 return result;
             })
        .def("modulate_batch",
             [](modulator_kernel_cc& self, const carray array) {
                 py::buffer_info in = array.request();
                 require_batch(in, self.block_size(), "Input");
                 auto result = like(in);
                 py::buffer_info out = result.request();
                 self.generic_work_batch(ptr(out), cptr(in), in.size / self.block_size());
                 return result;
             },
             "modulate any whole number of blocks (array of size nblocks*block_size, any shape) in one launch");

    py::class_<receiver_kernel_cc>(m, "Demodulator")
        .def(py::init<int, int, int, std::vector<cfloat>>(), py::arg("timeslots"), py::arg("subcarriers"), py::arg("overlap"),
             py::arg("frequency_taps"))
        .def("timeslots", &receiver_kernel_cc::timeslots)
        .def("subcarriers", &receiver_kernel_cc::subcarriers)
        .def("overlap", &receiver_kernel_cc::overlap)
        .def("block_size", &receiver_kernel_cc::block_size)
        .def("filter_taps", &receiver_kernel_cc::filter_taps)
        .def("ic_filter_taps", &receiver_kernel_cc::ic_filter_taps)
        .def("kernel_name", &receiver_kernel_cc::kernel_name)
        .def("demodulate", [](receiver_kernel_cc& self, const carray a) {
            return rx_unary(self, a, [&](cfloat* o, const cfloat* i) { self.generic_work(o, i); });
        })
        .def("fft_filter_downsample", [](receiver_kernel_cc& self, const carray a) {
            return rx_unary(self, a, [&](cfloat* o, const cfloat* i) { self.fft_filter_downsample(o, i); });
        })
        .def("transform_subcarriers_to_td", [](receiver_kernel_cc& self, const carray a) {
            return rx_unary(self, a, [&](cfloat* o, const cfloat* i) { self.transform_subcarriers_to_td(o, i); });
        })
        .def("demodulate_equalize", [](receiver_kernel_cc& self, const carray a, const carray eq) {
            return rx_binary(self, a, eq, [&](cfloat* o, const cfloat* i, const cfloat* e) { self.generic_work_equalize(o, i, e); });
        })
        .def("fft_equalize_filter_downsample", [](receiver_kernel_cc& self, const carray a, const carray eq) {
            return rx_binary(self, a, eq, [&](cfloat* o, const cfloat* i, const cfloat* e) { self.fft_equalize_filter_downsample(o, i, e); });
        })
        .def("cancel_sc_interference", [](receiver_kernel_cc& self, const carray td, const carray fd) {
            return rx_binary(self, td, fd, [&](cfloat* o, const cfloat* i, const cfloat* e) { self.cancel_sc_interference(o, i, e); });
        })
        .def("configure_frames",
             [](receiver_kernel_cc& self, int frame_len, int cp_len, std::vector<int> smap, bool per_timeslot) {
                 self.configure_frames(frame_len, cp_len, smap, per_timeslot);
             },
             py::arg("frame_len"), py::arg("cp_len"), py::arg("subcarrier_map") = std::vector<int>(), py::arg("per_timeslot") = true)
        .def("demodulate_frames",
             [](receiver_kernel_cc& self, const carray frames, py::object eq, int noutput_size) {
                 return run_frames(self, frames, eq, noutput_size);
             },
             py::arg("frames"), py::arg("f_eq") = py::none(), py::arg("noutput_size") = 0,
             "cyclic-prefix removal + demodulation + resource demapping of whole frames in one kernel launch")
        .def("set_channel_estimator",
             [](receiver_kernel_cc& self, preamble_channel_estimator_cc* est) {
                 self.set_channel_estimator(est);
             },
             py::arg("estimator").none(true), py::keep_alive<1, 2>())
        .def("demodulate_estimated",
             [](receiver_kernel_cc& self, const carray x, const carray rx_preamble, int preamble_stride, int noutput_size) {
                 return run_estimated(self, x, rx_preamble, preamble_stride, noutput_size);
             },
             py::arg("x"), py::arg("rx_preamble"), py::arg("preamble_stride") = 0, py::arg("noutput_size") = 0,
             "channel estimation from each block's received preamble + demodulation (+ prefix removal / demapping) in one kernel launch")
        .def("demodulate_batch",
             [](receiver_kernel_cc& self, const carray array, py::object eq) {
                 py::buffer_info in = array.request();
                 require_batch(in, self.block_size(), "Input");
                 auto result = like(in);
                 py::buffer_info out = result.request();
                 if (eq.is_none()) {
                     self.generic_work_batch(ptr(out), cptr(in), nullptr, in.size / self.block_size());
                 } else {
                     carray eq_arr = eq.cast<carray>();
                     py::buffer_info e = eq_arr.request();
                     if (e.size != in.size) throw std::runtime_error("Channel vector size(" + std::to_string(e.size) + ") MUST be equal to input size(" + std::to_string(in.size) + ")!");
                     self.generic_work_batch(ptr(out), cptr(in), cptr(e), in.size / self.block_size());
                 }
                 return result;
             },
             py::arg("frames"), py::arg("f_eq") = py::none(), "demodulate any whole number of blocks; f_eq holds one vector per block");

    py::class_<constellation, std::shared_ptr<constellation>>(m, "Constellation")
        .def(py::init([](std::vector<cfloat> points) { return constellation::from_points(std::move(points)); }))
        .def_static("qpsk", &constellation::qpsk)
        .def_static("bpsk", &constellation::bpsk)
        .def("points", &constellation::points)
        .def("decision_maker", [](const constellation& c, cfloat s) { return c.decision_maker(&s); });

    py::class_<advanced_receiver_kernel_cc>(m, "AdvancedReceiver")
        .def(py::init<int, int, int, std::vector<cfloat>, std::vector<int>, int, constellation_sptr, int>(), py::arg("timeslots"),
             py::arg("subcarriers"), py::arg("overlap"), py::arg("frequency_taps"), py::arg("subcarrier_map"), py::arg("ic_iter"),
             py::arg("constellation"), py::arg("do_phase_compensation") = 0)
        .def("block_size", &advanced_receiver_kernel_cc::block_size)
        .def("set_ic", &advanced_receiver_kernel_cc::set_ic)
        .def("get_ic", &advanced_receiver_kernel_cc::get_ic)
        .def("set_phase_compensation", &advanced_receiver_kernel_cc::set_phase_compensation)
        .def("get_phase_compensation", &advanced_receiver_kernel_cc::get_phase_compensation)
        .def("kernel_name", &advanced_receiver_kernel_cc::kernel_name)
        .def("configure_frames",
             [](advanced_receiver_kernel_cc& self, int frame_len, int cp_len, std::vector<int> smap, bool per_timeslot, int timeslots) {
                 (void)timeslots;
                 self.configure_frames(frame_len, cp_len, smap, per_timeslot);
             },
             py::arg("frame_len"), py::arg("cp_len"), py::arg("subcarrier_map"), py::arg("per_timeslot"), py::arg("timeslots"))
        .def("demodulate_frames",
             [](advanced_receiver_kernel_cc& self, const carray frames, py::object eq, int noutput_size) {
                 return run_frames(self, frames, eq, noutput_size);
             },
             py::arg("frames"), py::arg("f_eq") = py::none(), py::arg("noutput_size") = 0)
        .def("set_channel_estimator",
             [](advanced_receiver_kernel_cc& self, preamble_channel_estimator_cc* est) {
                 self.set_channel_estimator(est);
             },
             py::arg("estimator").none(true), py::keep_alive<1, 2>())
        .def("demodulate_estimated",
             [](advanced_receiver_kernel_cc& self, const carray x, const carray rx_preamble, int preamble_stride, int noutput_size) {
                 return run_estimated(self, x, rx_preamble, preamble_stride, noutput_size);
             },
             py::arg("x"), py::arg("rx_preamble"), py::arg("preamble_stride") = 0, py::arg("noutput_size") = 0)
        .def("demodulate",
             [](advanced_receiver_kernel_cc& self, const carray array) {
                 py::buffer_info in = array.request();
                 require_batch(in, self.block_size(), "Input");
                 auto result = like(in);
                 py::buffer_info out = result.request();
                 self.generic_work_batch(ptr(out), cptr(in), nullptr, in.size / self.block_size());
                 return result;
             })
        .def("demodulate_equalize", [](advanced_receiver_kernel_cc& self, const carray array, const carray eq) {
            py::buffer_info in = array.request(), e = eq.request();
            require_batch(in, self.block_size(), "Input");
            if (e.size != in.size) throw std::runtime_error("Channel vector size(" + std::to_string(e.size) + ") MUST be equal to input size(" + std::to_string(in.size) + ")!");
            auto result = like(in);
            py::buffer_info out = result.request();
            self.generic_work_batch(ptr(out), cptr(in), cptr(e), in.size / self.block_size());
            return result;
        });

    // composite transmitter kernel (the reference only binds the GNU Radio block transmitter_cc,
    // python/bindings/transmitter_cc_python.cc:36-70: same constructor argument order minus the length tag key)
    py::class_<transmitter_kernel>(m, "Transmitter")
        .def(py::init<int, int, int, int, int, int, std::vector<int>, bool, int, std::vector<cfloat>, std::vector<cfloat>, std::vector<int>,
                      std::vector<std::vector<cfloat>>>(),
             py::arg("timeslots"), py::arg("subcarriers"), py::arg("active_subcarriers"), py::arg("cp_len"), py::arg("cs_len"),
             py::arg("ramp_len"), py::arg("subcarrier_map"), py::arg("per_timeslot"), py::arg("overlap"), py::arg("frequency_taps"),
             py::arg("window_taps"), py::arg("cyclic_shifts"), py::arg("preambles"))
        .def("input_vector_size", &transmitter_kernel::input_vector_size)
        .def("output_vector_size", &transmitter_kernel::output_vector_size)
        .def("cyclic_shifts", &transmitter_kernel::cyclic_shifts)
        .def("kernel_name", &transmitter_kernel::kernel_name)
        .def("transmit",
             [](transmitter_kernel& self, const carray array, int n_ports) {
                 py::buffer_info in = array.request();
                 const int nin = self.input_vector_size();
                 if (in.size == 0 || in.size % nin)
                     throw std::runtime_error("Input size(" + std::to_string(in.size) + ") MUST be a multiple of input_vector_size(" +
                                              std::to_string(nin) + ")!");
                 const long nframes = in.size / nin;
                 if (n_ports <= 0) n_ports = static_cast<int>(self.cyclic_shifts().size());
                 std::vector<py::array_t<cfloat>> outs;
                 std::vector<cfloat*> ptrs;
                 for (int i = 0; i < n_ports; ++i) {
                     outs.emplace_back(std::vector<py::ssize_t>{ nframes, self.output_vector_size() });
                     ptrs.push_back(static_cast<cfloat*>(outs.back().request().ptr));
                 }
                 self.generic_work_batch(ptrs.data(), n_ports, cptr(in), nin, nframes);
                 return outs;
             },
             py::arg("symbols"), py::arg("n_ports") = 0,
             "frames of every cyclic shift (list of arrays [nframes, output_vector_size]) for nframes * input_vector_size symbols");

    // python/bindings/preamble_channel_estimator_python.cc:30-99 (same class name, constructor arguments, methods, messages);
    // estimate_frame / estimate_snr additionally accept [nframes][2 * subcarriers] batches, estimate_snr_cnrs returns the
    // per-subcarrier CNRs the reference computes but drops at the binding (:94-97)
    py::class_<preamble_channel_estimator_cc>(m, "Preamble_channel_estimator")
        .def(py::init<int, int, int, bool, int, std::vector<cfloat>>(), py::arg("timeslots"), py::arg("subcarriers"),
             py::arg("active_subcarriers"), py::arg("is_dc_free"), py::arg("which_estimator"), py::arg("preamble"))
        .def("timeslots", &preamble_channel_estimator_cc::timeslots)
        .def("subcarriers", &preamble_channel_estimator_cc::fft_len)
        .def("active_subcarriers", &preamble_channel_estimator_cc::active_subcarriers)
        .def("frame_len", &preamble_channel_estimator_cc::frame_len)
        .def("is_dc_free", &preamble_channel_estimator_cc::is_dc_free)
        .def("preamble_filter_taps", &preamble_channel_estimator_cc::preamble_filter_taps)
        .def("estimate_frame",
             [](preamble_channel_estimator_cc& self, const carray array) {
                 py::buffer_info in = array.request();
                 const py::ssize_t n = 2 * self.fft_len();
                 if (in.ndim > 2) throw std::runtime_error("Only ONE-dimensional vectors allowed!");
                 if ((in.ndim == 1 && in.size != n) || (in.ndim == 2 && in.shape[1] != n))
                     throw std::runtime_error("Input vector size(" + std::to_string(in.ndim == 2 ? in.shape[1] : in.size) +
                                              ") MUST be equal to 2 * subcarriers(" + std::to_string(n) + ")!");
                 const long nframes = in.ndim == 2 ? static_cast<long>(in.shape[0]) : 1;
                 auto result = in.ndim == 2 ? py::array_t<cfloat>(std::vector<py::ssize_t>{ nframes, self.frame_len() })
                                            : py::array_t<cfloat>(self.frame_len());
                 py::buffer_info out = result.request();
                 self.estimate_frame_batch(ptr(out), cptr(in), nframes);
                 return result;
             })
        .def("estimate_snr",
             [](preamble_channel_estimator_cc& self, const carray array) {
                 py::buffer_info in = array.request();
                 if (in.ndim != 1) throw std::runtime_error("Only ONE-dimensional vectors allowed!");
                 if (in.size != 2 * self.fft_len())
                     throw std::runtime_error("Input vector size(" + std::to_string(in.size) + ") MUST be equal to 2 * subcarriers(" +
                                              std::to_string(2 * self.fft_len()) + ")!");
                 std::vector<float> cnrs;
                 return self.estimate_snr(cnrs, cptr(in));
             })
        .def("estimate_snr_cnrs", [](preamble_channel_estimator_cc& self, const carray array) {
            py::buffer_info in = array.request();
            const py::ssize_t n = 2 * self.fft_len();
            if (in.size == 0 || in.size % n)
                throw std::runtime_error("Input size(" + std::to_string(in.size) + ") MUST be a multiple of 2 * subcarriers(" + std::to_string(n) + ")!");
            const long nframes = in.size / n;
            py::array_t<float> snr(nframes);
            py::array_t<float> cnrs(std::vector<py::ssize_t>{ nframes, self.active_subcarriers() });
            self.estimate_snr_batch(static_cast<float*>(snr.request().ptr), static_cast<float*>(cnrs.request().ptr), cptr(in), nframes);
            return py::make_tuple(snr, cnrs);
        });
    // python/bindings/resource_mapper_python.cc:30-87 (same class name, constructor arguments, methods, messages -- the reference's
    // messages say "Modulator.block_size" here too); both methods additionally accept [nblocks][size] batches
    py::class_<resource_mapper_kernel_cc>(m, "Resource_mapper")
        .def(py::init<int, int, int, std::vector<int>, bool>())
        .def(py::init<int, int, int, std::vector<int>, bool, bool>(), py::arg("timeslots"), py::arg("subcarriers"), py::arg("active_subcarriers"),
             py::arg("subcarrier_map"), py::arg("per_timeslot"), py::arg("is_mapper"))          // addition: the demapper blocks' direction flag
        .def("input_vector_size", &resource_mapper_kernel_cc::input_vector_size)
        .def("output_vector_size", &resource_mapper_kernel_cc::output_vector_size)
        .def("block_size", &resource_mapper_kernel_cc::block_size)
        .def("frame_size", &resource_mapper_kernel_cc::frame_size)
        .def("map_to_resources",
             [](resource_mapper_kernel_cc& self, const carray array) {
                 py::buffer_info in = array.request();
                 const py::ssize_t n = static_cast<py::ssize_t>(self.block_size()), f = static_cast<py::ssize_t>(self.frame_size());
                 if (in.ndim != 1 && in.ndim != 2) throw std::runtime_error("Only ONE-dimensional vectors allowed!");
                 const py::ssize_t got = in.ndim == 2 ? in.shape[1] : in.size;
                 if (got != n)
                     throw std::runtime_error("Input vector size(" + std::to_string(got) + ") MUST be equal to Modulator.block_size(" +
                                              std::to_string(n) + ")!");
                 const long nblocks = in.ndim == 2 ? static_cast<long>(in.shape[0]) : 1;
                 auto result = in.ndim == 2 ? py::array_t<cfloat>(std::vector<py::ssize_t>{ nblocks, f }) : py::array_t<cfloat>(f);
                 py::buffer_info out = result.request();
                 self.map_to_resources_batch(ptr(out), cptr(in), self.block_size(), nblocks);
                 return result;
             })
        .def("demap_from_resources", [](resource_mapper_kernel_cc& self, const carray array) {
            py::buffer_info in = array.request();
            const py::ssize_t n = static_cast<py::ssize_t>(self.block_size()), f = static_cast<py::ssize_t>(self.frame_size());
            if (in.ndim != 1 && in.ndim != 2) throw std::runtime_error("Only ONE-dimensional vectors allowed!");
            const py::ssize_t got = in.ndim == 2 ? in.shape[1] : in.size;
            if (got != f)
                throw std::runtime_error("Input vector size(" + std::to_string(got) + ") MUST be equal to Modulator.block_size(" + std::to_string(f) +
                                         ")!");
            const long nblocks = in.ndim == 2 ? static_cast<long>(in.shape[0]) : 1;
            auto result = in.ndim == 2 ? py::array_t<cfloat>(std::vector<py::ssize_t>{ nblocks, n }) : py::array_t<cfloat>(n);
            py::buffer_info out = result.request();
            self.demap_from_resources_batch(ptr(out), cptr(in), self.block_size(), nblocks);
            return result;
        });

    // python/bindings/cyclic_prefix_python.cc:31-93 (same class name, keyword arguments, methods, messages); both methods additionally
    // accept [nblocks][size] batches
    py::class_<add_cyclic_prefix_cc>(m, "Cyclic_prefixer")
        .def(py::init<int, int, int, int, std::vector<cfloat>, int>(), py::arg("block_len"), py::arg("cp_len"), py::arg("cs_len"),
             py::arg("ramp_len"), py::arg("window_taps"), py::arg("cyclic_shift") = 0)
        .def("block_size", &add_cyclic_prefix_cc::block_size)
        .def("frame_size", &add_cyclic_prefix_cc::frame_size)
        .def("cyclic_shift", &add_cyclic_prefix_cc::cyclic_shift)
        .def("add_cyclic_prefix",
             [](add_cyclic_prefix_cc& self, const carray array) {
                 py::buffer_info in = array.request();
                 const py::ssize_t n = self.block_size(), f = self.frame_size();
                 if (in.ndim != 1 && in.ndim != 2) throw std::runtime_error("Only ONE-dimensional vectors allowed!");
                 const py::ssize_t got = in.ndim == 2 ? in.shape[1] : in.size;
                 if (got != n)
                     throw std::runtime_error("Input vector size(" + std::to_string(got) + ") MUST be equal to Cyclic_prefix.block_size(" +
                                              std::to_string(n) + ")!");
                 const long nblocks = in.ndim == 2 ? static_cast<long>(in.shape[0]) : 1;
                 auto result = in.ndim == 2 ? py::array_t<cfloat>(std::vector<py::ssize_t>{ nblocks, f }) : py::array_t<cfloat>(f);
                 py::buffer_info out = result.request();
                 self.add_cyclic_prefix_batch(ptr(out), cptr(in), self.cyclic_shift(), nblocks);
                 return result;
             })
        .def("remove_cyclic_prefix", [](add_cyclic_prefix_cc& self, const carray array) {
            py::buffer_info in = array.request();
            const py::ssize_t n = self.block_size(), f = self.frame_size();
            if (in.ndim != 1 && in.ndim != 2) throw std::runtime_error("Only ONE-dimensional vectors allowed!");
            const py::ssize_t got = in.ndim == 2 ? in.shape[1] : in.size;
            if (got != f)                       // the reference prints block_size() in this message (cyclic_prefix_python.cc:80-84)
                throw std::runtime_error("Input vector size(" + std::to_string(got) + ") MUST be equal to Cyclic_prefix.frame_size(" +
                                         std::to_string(n) + ")!");
            const long nblocks = in.ndim == 2 ? static_cast<long>(in.shape[0]) : 1;
            auto result = in.ndim == 2 ? py::array_t<cfloat>(std::vector<py::ssize_t>{ nblocks, n }) : py::array_t<cfloat>(n);
            py::buffer_info out = result.request();
            self.remove_cyclic_prefix_batch(ptr(out), cptr(in), nblocks);
            return result;
        });
}
