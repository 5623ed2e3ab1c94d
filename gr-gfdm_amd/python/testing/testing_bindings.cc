// gfdm_testing: a TEST-ONLY module (built next to gfdm_python, never imported by the product) with hooks the tests use to drive
// C++-only surfaces of the drop-in classes.  Its functions take the kernel objects of gfdm_python (imported first).
//   * a scheduler stand-in for the batched work() bodies of gfdm/batched_work.h: it calls them the way GNU Radio's scheduler
//     calls a block's work() -- successive runs of `noutput_items`, pointers advanced by what work() returned
//   * the reference's legacy 2-D receiver API (lib/receiver_kernel_cc.cc:130-163,194-209,227-272), which has no Python binding
//     in the reference either, and gfdm_kernel_utils::calculate_signal_energy (lib/gfdm_kernel_utils.cc:59-65)
//   * gfdm/sharded_batch.h (one batch over several devices), which is a C++ template with no Python surface of its own
#include <pybind11/complex.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdlib>
#include <cstring>
#include <memory>

#include <gfdm/add_cyclic_prefix_cc.h>
#include <gfdm/advanced_receiver_kernel_cc.h>
#include <gfdm/batched_work.h>
#include <gfdm/host_memory.h>
#include <gfdm/modulator_kernel_cc.h>
#include <gfdm/preamble_channel_estimator_cc.h>
#include <gfdm/receiver_kernel_cc.h>
#include <gfdm/resource_mapper_kernel_cc.h>
#include <gfdm/sharded_batch.h>
#include <gfdm/transmitter_kernel.h>

namespace py = pybind11;
using namespace gr::gfdm;

typedef std::complex<float> cfloat;
typedef py::array_t<cfloat, py::array::c_style | py::array::forcecast> carray;

namespace {

// work(n) for n in chunks; in / out advance by the returned item count (sync block: consumed == produced)
// registered: the two streams play the scheduler's long-lived buffers, pinned once with gr::gfdm::host_registration (gfdm/host_memory.h)
template <class Kernel>
py::tuple run_sync(Kernel& k, const carray& in_arr, const std::vector<int>& chunks, bool registered)
{
    py::buffer_info in = in_arr.request();
    py::array_t<cfloat> out_arr(in.size);
    cfloat* out = static_cast<cfloat*>(out_arr.request().ptr);
    std::fill(out, out + in.size, cfloat(0.f, 0.f));
    const cfloat* src = static_cast<const cfloat*>(in.ptr);
    // the scheduler's buffers: whole pages (as GNU Radio's mmap'ed circular buffers are), registered for the lifetime of the run
    std::unique_ptr<host_registration> reg_in, reg_out;
    struct Free { void operator()(void* p) const { free(p); } };
    std::unique_ptr<void, Free> ring_in, ring_out;
    cfloat* user_out = out;
    if (registered && in.size > 0) {
        const size_t bytes = ((size_t)in.size * sizeof(cfloat) + 4095) / 4096 * 4096;
        ring_in.reset(aligned_alloc(4096, bytes));
        ring_out.reset(aligned_alloc(4096, bytes));
        if (!ring_in || !ring_out) throw std::bad_alloc();
        memcpy(ring_in.get(), src, (size_t)in.size * sizeof(cfloat));
        memset(ring_out.get(), 0, bytes);
        reg_in = std::make_unique<host_registration>(ring_in.get(), bytes);
        reg_out = std::make_unique<host_registration>(ring_out.get(), bytes);
        src = static_cast<const cfloat*>(ring_in.get());
        out = static_cast<cfloat*>(ring_out.get());
    }
    std::vector<int> returned;
    long pos = 0;
    for (int n : chunks) {
        if (pos + n > in.size) throw std::runtime_error("scheduler stand-in: chunk runs past the input");
        const int r = batched::sync_work(k, n, src + pos, out + pos);
        returned.push_back(r);
        pos += r;
    }
    if (out != user_out) memcpy(user_out, out, (size_t)in.size * sizeof(cfloat));
    return py::make_tuple(out_arr, returned);
}

py::tuple run_sync_eq(advanced_receiver_kernel_cc& k, const carray& in_arr, py::object eq_obj, const std::vector<int>& chunks)
{
    py::buffer_info in = in_arr.request();
    carray eq_arr;
    const cfloat* eq = nullptr;
    if (!eq_obj.is_none()) {
        eq_arr = eq_obj.cast<carray>();
        if (eq_arr.request().size != in.size) throw std::runtime_error("equaliser stream must be as long as the input stream");
        eq = static_cast<const cfloat*>(eq_arr.request().ptr);
    }
    py::array_t<cfloat> out_arr(in.size);
    cfloat* out = static_cast<cfloat*>(out_arr.request().ptr);
    std::fill(out, out + in.size, cfloat(0.f, 0.f));
    const cfloat* src = static_cast<const cfloat*>(in.ptr);
    std::vector<int> returned;
    long pos = 0;
    for (int n : chunks) {
        if (pos + n > in.size) throw std::runtime_error("scheduler stand-in: chunk runs past the input");
        const int r = batched::sync_work_equalize(k, n, src + pos, eq ? eq + pos : nullptr, out + pos);
        returned.push_back(r);
        pos += r;
    }
    return py::make_tuple(out_arr, returned);
}

typedef std::vector<std::vector<cfloat>> matrix_t;

matrix_t to_matrix(const carray& a, int K, int M)
{
    py::buffer_info b = a.request();
    if (b.size != static_cast<py::ssize_t>(K) * M) throw std::runtime_error("expected a [subcarriers][timeslots] array");
    const cfloat* p = static_cast<const cfloat*>(b.ptr);
    matrix_t m(K, std::vector<cfloat>(M));
    for (int k = 0; k < K; ++k) std::copy_n(p + static_cast<size_t>(k) * M, M, m[k].begin());
    return m;
}

py::array_t<cfloat> from_matrix(const matrix_t& m)
{
    const py::ssize_t K = m.size(), M = m.empty() ? 0 : m[0].size();
    py::array_t<cfloat> a(std::vector<py::ssize_t>{ K, M });
    cfloat* p = static_cast<cfloat*>(a.request().ptr);
    for (py::ssize_t k = 0; k < K; ++k) std::copy_n(m[k].begin(), M, p + k * M);
    return a;
}

} // namespace

PYBIND11_MODULE(gfdm_testing, t)
{
    t.doc() = "test hooks: scheduler stand-in for gfdm/batched_work.h, legacy 2-D receiver API, gfdm/sharded_batch.h";
    py::module_::import("gfdm_python");                  // the kernel classes the functions below take as arguments

    // gfdm/sharded_batch.h: one host batch over `devices` (an ordinal may repeat: several handles on one GPU); returns (result, shards)
    t.def("sharded_modulate", [](int M, int K, int L, const std::vector<cfloat>& taps, const std::vector<int>& devices, const carray& in_arr) {
        sharded_batch<modulator_kernel_cc> sb(devices, M, K, L, taps);
        py::buffer_info in = in_arr.request();
        const long nblocks = in.size / sb.block_size();
        py::array_t<cfloat> out(in.size);
        {
            py::gil_scoped_release nogil;
            sb.generic_work_batch(static_cast<cfloat*>(out.request().ptr), static_cast<const cfloat*>(in.ptr), nblocks);
        }
        std::vector<std::pair<long, long>> shards;
        for (int i = 0; i < sb.n_shards(); ++i) shards.push_back(sb.shard(nblocks, i));
        return py::make_tuple(out, shards);
    });
    t.def("sharded_advanced_receive", [](int M, int K, int L, const std::vector<cfloat>& taps, const std::vector<int>& smap, int ic_iter,
                                         const std::vector<int>& devices, const carray& in_arr, py::object eq_obj) {
        sharded_batch<advanced_receiver_kernel_cc> sb(devices, M, K, L, taps, smap, ic_iter, constellation::qpsk(), 0);
        py::buffer_info in = in_arr.request();
        const long nblocks = in.size / sb.block_size();
        carray eq_arr;
        const cfloat* eq = nullptr;
        if (!eq_obj.is_none()) { eq_arr = eq_obj.cast<carray>(); eq = static_cast<const cfloat*>(eq_arr.request().ptr); }
        py::array_t<cfloat> out(in.size);
        {
            py::gil_scoped_release nogil;
            sb.generic_work_batch(static_cast<cfloat*>(out.request().ptr), static_cast<const cfloat*>(in.ptr), eq, nblocks);
        }
        return out;
    });
    t.def("shard_range", [](long total, int index, int n) { return shard_range(total, index, n); });
    t.def("default_device_is_per_thread", []() {
        const int before = gfdm_kernel_utils::set_default_device(0);
        int seen_in_thread = -1;
        gfdm_kernel_utils::set_default_device(3);
        std::thread th([&] { seen_in_thread = gfdm_kernel_utils::default_device(); });
        th.join();
        const bool ok = (seen_in_thread == 0) && gfdm_kernel_utils::default_device() == 3;
        gfdm_kernel_utils::set_default_device(before);
        return ok;
    });
    t.def("scheduler_run", &run_sync<modulator_kernel_cc>, py::arg("kernel"), py::arg("stream"), py::arg("noutput_items"), py::arg("registered") = false);
    t.def("scheduler_run", &run_sync<receiver_kernel_cc>, py::arg("kernel"), py::arg("stream"), py::arg("noutput_items"), py::arg("registered") = false);
    t.def("scheduler_run_equalize", &run_sync_eq, py::arg("kernel"), py::arg("stream"), py::arg("eq_stream"), py::arg("noutput_items"));
    t.def("scheduler_run_transmitter",
          [](transmitter_kernel& k, const carray& in_arr, const std::vector<std::pair<int, int>>& calls, int n_ports) {
              py::buffer_info in = in_arr.request();
              const int nin = k.input_vector_size(), nout = k.output_vector_size();
              const long max_frames = in.size / nin;
              std::vector<py::array_t<cfloat>> outs;
              std::vector<cfloat*> base;
              for (int p = 0; p < n_ports; ++p) {
                  outs.emplace_back(std::vector<py::ssize_t>{ max_frames, nout });
                  base.push_back(static_cast<cfloat*>(outs.back().request().ptr));
                  std::fill(base.back(), base.back() + max_frames * nout, cfloat(0.f, 0.f));
              }
              const cfloat* src = static_cast<const cfloat*>(in.ptr);
              std::vector<int> frames;
              long done = 0;
              for (auto& c : calls) {                       // (noutput_items, ninput_items[0]) of one general_work call
                  if (done * nin + c.second > in.size) throw std::runtime_error("scheduler stand-in: call runs past the input");
                  std::vector<cfloat*> ptrs;
                  for (int p = 0; p < n_ports; ++p) ptrs.push_back(base[p] + done * nout);
                  const int f = batched::transmitter_work(k, c.first, c.second, src + done * nin, ptrs.data(), n_ports);
                  frames.push_back(f);
                  done += f;
              }
              return py::make_tuple(outs, frames);
          },
          py::arg("kernel"), py::arg("symbols"), py::arg("calls"), py::arg("n_ports"));
    // fixed-rate general blocks with one input and one output: calls = (noutput_items, ninput_items[0]) per general_work call; returns the
    // output stream, the frames each call produced
    t.def("scheduler_run_mapper",
          [](resource_mapper_kernel_cc& k, bool is_mapper, const carray& in_arr, const std::vector<std::pair<int, int>>& calls) {
              py::buffer_info in = in_arr.request();
              const long nin = static_cast<long>(k.input_vector_size()), nout = static_cast<long>(k.output_vector_size());
              const long max_frames = in.size / nin;
              py::array_t<cfloat> out_arr(std::vector<py::ssize_t>{ max_frames, nout });
              cfloat* out = static_cast<cfloat*>(out_arr.request().ptr);
              std::fill(out, out + max_frames * nout, cfloat(0.f, 0.f));
              const cfloat* src = static_cast<const cfloat*>(in.ptr);
              std::vector<int> frames;
              long done = 0;
              for (auto& c : calls) {
                  if (done * nin + c.second > in.size) throw std::runtime_error("scheduler stand-in: call runs past the input");
                  const int f = batched::mapper_work(k, is_mapper, c.first, c.second, src + done * nin, out + done * nout);
                  frames.push_back(f);
                  done += f;
              }
              return py::make_tuple(out_arr, frames);
          },
          py::arg("kernel"), py::arg("is_mapper"), py::arg("stream"), py::arg("calls"));
    t.def("scheduler_run_prefixer",
          [](add_cyclic_prefix_cc& k, const carray& in_arr, const std::vector<int>& chunks) {
              py::buffer_info in = in_arr.request();
              const long nin = k.block_size(), nout = k.frame_size();
              const long max_frames = in.size / nin;
              py::array_t<cfloat> out_arr(std::vector<py::ssize_t>{ max_frames, nout });
              cfloat* out = static_cast<cfloat*>(out_arr.request().ptr);
              std::fill(out, out + max_frames * nout, cfloat(0.f, 0.f));
              const cfloat* src = static_cast<const cfloat*>(in.ptr);
              std::vector<int> frames;
              long done = 0;
              for (int n : chunks) {
                  if (done + n / nout > max_frames) throw std::runtime_error("scheduler stand-in: chunk runs past the input");
                  const int f = batched::prefixer_work(k, n, src + done * nin, out + done * nout);
                  frames.push_back(f);
                  done += f;
              }
              return py::make_tuple(out_arr, frames);
          },
          py::arg("kernel"), py::arg("stream"), py::arg("noutput_items"));
    t.def("scheduler_run_estimator",
          [](preamble_channel_estimator_cc& k, const carray& in_arr, const std::vector<int>& chunks) {
              py::buffer_info in = in_arr.request();
              const int pre_len = 2 * k.fft_len(), frame_len = k.frame_len();
              const long max_frames = in.size / pre_len;
              py::array_t<cfloat> out_arr(std::vector<py::ssize_t>{ max_frames, frame_len });
              cfloat* out = static_cast<cfloat*>(out_arr.request().ptr);
              std::fill(out, out + max_frames * frame_len, cfloat(0.f, 0.f));
              const cfloat* src = static_cast<const cfloat*>(in.ptr);
              py::list tags;
              std::vector<int> frames;
              long done = 0;
              for (int n : chunks) {
                  if ((done + n / frame_len) > max_frames) throw std::runtime_error("scheduler stand-in: chunk runs past the input");
                  const int f = batched::estimator_work(k, n, src + done * pre_len, out + done * frame_len,
                                                        [&](int i, float snr, const float* cnrs, int ncnrs) {
                                                            tags.append(py::make_tuple(done + i, snr, std::vector<float>(cnrs, cnrs + ncnrs)));
                                                        });
                  frames.push_back(f);
                  done += f;
              }
              return py::make_tuple(out_arr, frames, tags);
          },
          py::arg("kernel"), py::arg("rx_preambles"), py::arg("noutput_items"));

    // legacy 2-D API of receiver_kernel_cc
    t.def("legacy_filter_superposition", [](receiver_kernel_cc& k, const carray& frame) {
        py::buffer_info in = frame.request();
        if (in.size != k.block_size()) throw std::runtime_error("expected one block");
        matrix_t out(k.subcarriers(), std::vector<cfloat>(k.timeslots()));
        k.filter_superposition(out, static_cast<const cfloat*>(in.ptr));
        return from_matrix(out);
    });
    t.def("legacy_demodulate_subcarrier", [](receiver_kernel_cc& k, const carray& fd) {
        matrix_t in = to_matrix(fd, k.subcarriers(), k.timeslots()), out(k.subcarriers(), std::vector<cfloat>(k.timeslots()));
        k.demodulate_subcarrier(out, in);
        return from_matrix(out);
    });
    t.def("legacy_remove_sc_interference", [](receiver_kernel_cc& k, const carray& symbols, const carray& fd) {
        matrix_t sym = to_matrix(symbols, k.subcarriers(), k.timeslots()), f = to_matrix(fd, k.subcarriers(), k.timeslots());
        k.remove_sc_interference(sym, f);                    // result replaces sc_symbols, as in the reference
        return from_matrix(sym);
    });
    t.def("legacy_vectorize_serialize", [](receiver_kernel_cc& k, const carray& flat) {
        py::buffer_info in = flat.request();
        if (in.size != k.block_size()) throw std::runtime_error("expected one block");
        matrix_t mat(k.subcarriers(), std::vector<cfloat>(k.timeslots()));
        k.vectorize_2d(mat, static_cast<const cfloat*>(in.ptr));
        py::array_t<cfloat> back(in.size);
        k.serialize_output(static_cast<cfloat*>(back.request().ptr), mat);
        return py::make_tuple(from_matrix(mat), back);
    });
    t.def("calculate_signal_energy", [](receiver_kernel_cc& k, const carray& x) {
        py::buffer_info in = x.request();
        return k.calculate_signal_energy(static_cast<const cfloat*>(in.ptr), static_cast<int>(in.size));
    });
    t.def("calculate_signal_energy", [](modulator_kernel_cc& k, const carray& x) {
        py::buffer_info in = x.request();
        return k.calculate_signal_energy(static_cast<const cfloat*>(in.ptr), static_cast<int>(in.size));
    });
}
