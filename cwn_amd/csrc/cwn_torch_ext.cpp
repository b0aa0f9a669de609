// cwn_torch_ext.cpp -- the compiled binding of the EAGER path (round 5).
//
// Why: replayed from a hipGraph a forward costs its kernels; launched eagerly -- what every caller outside the static-batch
// path does, layer by layer through the reference-shaped API (mp/layers.py:184-199: what a user layer calls per forward) --
// it cost 2.7 x (propagate scope) to 8 x (full forward) that, all of it host time: ctypes field stores, torch.empty /
// split from Python, per-tensor checks.  A prepared launch (cwn_amd/ops.py: LayerLaunch, MlpLaunch) is a descriptor array
// filled in ONCE from a layer's parameters and a batch's index tensors; per call only the feature / output pointers change.
// This module keeps a byte copy of that array (the structs of include/cwn_hip.h -- the C ABI stays the one boundary; the
// entry points are called through their addresses, handed over by the ctypes binding, so this module links against
// nothing of the library) and does the per-call part in C++: argument checks, ONE allocation for all outputs, the pointer
// patches, the current stream, the call.  ctypes stays the binding of everything that is not per-call work, and the
// fallback when this module is not built (CWN_BINDING=ctypes forces it: the parity tests run through both).
//
// Also registered as torch.library ops (cwn::layer_fused, cwn::update_mlp over a handle number) so that the launches are
// visible to the dispatcher (profilers, torch.ops.*); the Python fast path calls the methods directly.
//
// Built by cwn_amd/_build_ext.py (plain g++ against the torch headers; no device code in here).
#include <torch/extension.h>
#include <torch/library.h>
#include <c10/hip/HIPStream.h>

#include <cstring>
#include <mutex>
#include <optional>
#include <unordered_map>
#include <vector>

#include "../../include/cwn_hip.h"

namespace {

namespace py = pybind11;

using fused_fn = int (*)(const cwn_layer_dim*, int, int32_t, const cwn_layer_plan*, int32_t, int32_t*, cwn_stream_t);
using mlp_fn = int (*)(const cwn_mlp_dim*, int, int32_t, cwn_stream_t);

// a non-zero return code of the library becomes the exception the ctypes binding raises (cwn_amd._ffi.check -> CwnError)
[[noreturn]] void raise_rc(int rc, const char* what) {
    py::gil_scoped_acquire gil;
    py::module_::import("cwn_amd._ffi").attr("check")(rc, what);
    throw std::runtime_error(std::string(what) + ": error code " + std::to_string(rc));
}

inline void check_feature(const at::Tensor& x, int64_t rows, int64_t F, int dev, const char* who) {
    if (x.dim() != 2 || x.size(0) != rows || x.size(1) != F)
        throw py::value_error(std::string(who) + ": feature rows / width do not match the batch this launch was prepared for");
    if (x.scalar_type() != at::kFloat || !x.device().is_cuda() || x.device().index() != dev)
        throw py::type_error(std::string(who) + ": features must be float32 tensors on the GPU this launch was prepared for");
}

inline at::Tensor row_major(const at::Tensor& t) {      // ops._rowmajor
    if ((t.size(1) > 1 && t.stride(1) != 1) || (t.size(0) > 1 && t.stride(0) < t.size(1))) return t.contiguous();
    return t;
}

// ---- tensors a prepared record was derived from: unchanged since? ------------------------------------------------------------
class TensorMarks {
 public:
    // An inference tensor keeps no version counter (at::Tensor::_version() throws) and may be written in place inside
    // torch.inference_mode(): marks over one are never current, the caller prepares its launch again (_ffi.tver's rule).
    explicit TensorMarks(std::vector<at::Tensor> ts) : ts_(std::move(ts)) {
        marks_.reserve(ts_.size());
        for (const auto& t : ts_) {
            if (t.is_inference()) { untracked_ = true; marks_.emplace_back(t.data_ptr(), -1); continue; }
            marks_.emplace_back(t.data_ptr(), static_cast<int64_t>(t._version()));
        }
    }
    bool current() const {
        if (untracked_) return false;
        for (size_t i = 0; i < ts_.size(); ++i)
            if (ts_[i].is_inference() || static_cast<int64_t>(ts_[i]._version()) != marks_[i].second ||
                ts_[i].data_ptr() != marks_[i].first) return false;
        return true;
    }
    size_t size() const { return ts_.size(); }

 private:
    std::vector<at::Tensor> ts_;
    std::vector<std::pair<void*, int64_t>> marks_;
    bool untracked_ = false;
};

// ---- cwn_layer_fused_f32 over a prepared descriptor array (ops.LayerLaunch) --------------------------------------------------
// (A prepared launch patches ITS copy of the array per call: one caller at a time, like its ctypes form -- two streams that
// want the same layer concurrently prepare two launches.  A handle from layer_register / mlp_register keeps the C++ object alive,
// not the tensors its descriptors point at: those belong to the Python launch object.)
class LayerCall {
 public:
    // n_out: outputs per dimension -- 2 (out_up, out_b) or 3 (out_up, out_down, out_b: a CIN++ layer, cwn_layer_dim.out_down)
    LayerCall(uintptr_t arr, int64_t arr_bytes, int n, int F, std::vector<int64_t> rows, uintptr_t err, uintptr_t fn, int dev,
              int n_out)
        : n_(n), F_(F), dev_(dev), n_out_(n_out), rows_(std::move(rows)), err_(reinterpret_cast<int32_t*>(err)),
          fn_(reinterpret_cast<fused_fn>(fn)) {
        if (n_out != 2 && n_out != 3) throw py::value_error("LayerCall: two or three outputs per dimension");
        if (n < 1 || n > CWN_LAYER_MAX_DIMS || (int64_t)rows_.size() != n || arr_bytes != (int64_t)(n * sizeof(cwn_layer_dim)) || fn == 0)
            throw py::value_error("LayerCall: descriptor array does not match include/cwn_hip.h (ABI mismatch?)");
        dims_.resize(n);
        std::memcpy(dims_.data(), reinterpret_cast<const void*>(arr), (size_t)arr_bytes);
        total_ = 0;
        for (int d = 0; d < n; ++d) {
            big_y_.emplace_back(dims_[d].big_y1, dims_[d].big_y2);
            for (int k = 0; k < n_out_; ++k) sizes_.push_back(rows_[d]);
            total_ += rows_[d];
        }
    }
    bool has_plans(bool cached) const { return !plans_[cached ? 1 : 0].empty(); }
    void set_plans(bool cached, const std::vector<uintptr_t>& addrs, int64_t bytes_each) {
        if (bytes_each != (int64_t)sizeof(cwn_layer_plan)) throw py::value_error("LayerCall: cwn_layer_plan size mismatch");
        auto& v = plans_[cached ? 1 : 0];
        v.resize(addrs.size());
        for (size_t i = 0; i < addrs.size(); ++i) std::memcpy(&v[i], reinterpret_cast<const void*>(addrs[i]), sizeof(cwn_layer_plan));
    }
    using YPair = std::pair<std::optional<at::Tensor>, std::optional<at::Tensor>>;
    std::vector<at::Tensor> run(const std::vector<at::Tensor>& xs, int csr_mode, const std::optional<std::vector<YPair>>& ys) {
        if ((int)xs.size() != n_ || (ys && (int)ys->size() != n_)) throw py::value_error("LayerCall.run: one feature tensor per dimension");
        if (ys) csr_mode |= CWN_LAYER_STORE_Y;
        const bool cached = (csr_mode & (CWN_LAYER_CSR_STORE | CWN_LAYER_CSR_LOAD)) != 0;
        const auto& plans = plans_[cached ? 1 : 0];
        if (plans.empty()) throw py::value_error("LayerCall.run: plans of this form were not set");
        std::vector<at::Tensor> hold;
        hold.reserve(n_);
        for (int d = 0; d < n_; ++d) {
            check_feature(xs[d], rows_[d], F_, dev_, "cwn_layer_fused_f32");
            hold.push_back(xs[d].is_contiguous() ? xs[d] : xs[d].contiguous());
        }
        at::Tensor buf = at::empty({n_out_ * total_, (int64_t)F_}, at::TensorOptions().dtype(at::kFloat).device(at::kCUDA, dev_));
        float* base = buf.data_ptr<float>();
        int64_t off = 0;
        for (int d = 0; d < n_; ++d) {
            cwn_layer_dim& a = dims_[d];
            a.x = hold[d].data_ptr<float>();
            a.out_up = base + off * F_;
            a.out_b = base + (off + (n_out_ - 1) * rows_[d]) * F_;
            a.out_down = n_out_ == 3 ? base + (off + rows_[d]) * F_ : nullptr;
            off += n_out_ * rows_[d];
            a.big_y1 = big_y_[d].first;
            a.big_y2 = big_y_[d].second;
            if (ys) {
                const auto& y = (*ys)[d];
                if (y.first) a.big_y1 = y.first->data_ptr<float>();
                if (y.second) a.big_y2 = y.second->data_ptr<float>();
            }
        }
        cwn_stream_t stream = (cwn_stream_t)c10::hip::getCurrentHIPStream((c10::DeviceIndex)dev_).stream();
        for (const auto& plan : plans) {
            const int rc = fn_(dims_.data(), n_, F_, &plan, csr_mode, err_, stream);
            if (rc != 0) raise_rc(rc, "cwn_layer_fused_f32");
        }
        return buf.split_with_sizes(sizes_, 0);
    }

 private:
    int n_, F_, dev_, n_out_;
    std::vector<int64_t> rows_, sizes_;
    int64_t total_;
    int32_t* err_;
    fused_fn fn_;
    std::vector<cwn_layer_dim> dims_;
    std::vector<std::pair<float*, float*>> big_y_;
    std::vector<cwn_layer_plan> plans_[2];
};

// ---- cwn_update_mlp_f32 over a prepared descriptor array (ops.MlpLaunch) ------------------------------------------------------
class MlpCall {
 public:
    MlpCall(uintptr_t arr, int64_t arr_bytes, int n, int F, int64_t cap, uintptr_t fn, int dev, std::vector<at::Tensor> sources,
            int64_t state_epoch, int64_t struct_epoch)
        : n_(n), F_(F), dev_(dev), cap_(cap), fn_(reinterpret_cast<mlp_fn>(fn)), marks_(std::move(sources)), state_epoch_(state_epoch),
          struct_epoch_(struct_epoch) {
        if (n < 1 || n > CWN_LAYER_MAX_DIMS || arr_bytes != (int64_t)(n * sizeof(cwn_mlp_dim)) || fn == 0)
            throw py::value_error("MlpCall: descriptor array does not match include/cwn_hip.h (ABI mismatch?)");
        dims_.resize(n);
        std::memcpy(dims_.data(), reinterpret_cast<const void*>(arr), (size_t)arr_bytes);
    }
    bool current(int64_t state_epoch, int64_t struct_epoch) const {
        return state_epoch == state_epoch_ && struct_epoch == struct_epoch_ && marks_.current();
    }
    // None: these inputs are not what the launch takes (the caller goes the long way, which raises where something is wrong)
    py::object run(const std::vector<at::Tensor>& xs_up, const std::vector<at::Tensor>& xs_b) {
        if ((int)xs_up.size() != n_ || (int)xs_b.size() != n_) return py::none();
        std::vector<int64_t> rows(n_);
        int64_t total = 0;
        for (int i = 0; i < n_; ++i) {
            const at::Tensor &u = xs_up[i], &b = xs_b[i];
            if (u.dim() != 2 || b.dim() != 2) return py::none();
            const int64_t M = u.size(0);
            const int64_t w = dims_[i].in_width > 0 && dims_[i].in_width < F_ ? dims_[i].in_width : F_;    // (narrow inputs)
            if (M > cap_ || b.size(0) != M || u.size(1) != w || b.size(1) != w || u.scalar_type() != at::kFloat ||
                b.scalar_type() != at::kFloat || !u.device().is_cuda() || u.device().index() != dev_ || b.device() != u.device())
                return py::none();
            rows[i] = M;
            total += M;
        }
        at::Tensor buf = at::empty({total, (int64_t)F_}, at::TensorOptions().dtype(at::kFloat).device(at::kCUDA, dev_));
        float* base = buf.data_ptr<float>();
        std::vector<at::Tensor> hold;
        hold.reserve(2 * n_);
        int64_t off = 0;
        for (int i = 0; i < n_; ++i) {
            hold.push_back(row_major(xs_up[i]));
            hold.push_back(row_major(xs_b[i]));
            const at::Tensor &u = hold[2 * i], &b = hold[2 * i + 1];
            cwn_mlp_dim& a = dims_[i];
            a.x_up = u.data_ptr<float>();
            a.x_b = b.data_ptr<float>();
            a.y = base + off * F_;
            a.M = rows[i];
            a.ldx_up = rows[i] > 1 ? u.stride(0) : u.size(1);
            a.ldx_b = rows[i] > 1 ? b.stride(0) : b.size(1);
            a.m_dev = nullptr;                   // (capacity-sized launches -- _ffi.DYN_ROWS -- stay on the ctypes path)
            off += rows[i];
        }
        cwn_stream_t stream = (cwn_stream_t)c10::hip::getCurrentHIPStream((c10::DeviceIndex)dev_).stream();
        const int rc = fn_(dims_.data(), n_, F_, stream);
        if (rc != 0) raise_rc(rc, "cwn_update_mlp_f32");
        return py::cast(buf.split_with_sizes(rows, 0));
    }

 private:
    int n_, F_, dev_;
    int64_t cap_;
    mlp_fn fn_;
    TensorMarks marks_;
    int64_t state_epoch_, struct_epoch_;
    std::vector<cwn_mlp_dim> dims_;
};

// ---- handles for the torch.library form --------------------------------------------------------------------------------------
std::mutex g_mu;
std::unordered_map<int64_t, std::shared_ptr<LayerCall>> g_layer;
std::unordered_map<int64_t, std::shared_ptr<MlpCall>> g_mlp;
int64_t g_next = 1;

int64_t layer_register(std::shared_ptr<LayerCall> c) {
    std::lock_guard<std::mutex> l(g_mu);
    g_layer[g_next] = std::move(c);
    return g_next++;
}
int64_t mlp_register(std::shared_ptr<MlpCall> c) {
    std::lock_guard<std::mutex> l(g_mu);
    g_mlp[g_next] = std::move(c);
    return g_next++;
}
void release(int64_t h) {
    std::lock_guard<std::mutex> l(g_mu);
    g_layer.erase(h);
    g_mlp.erase(h);
}

std::vector<at::Tensor> op_layer_fused(at::TensorList xs, int64_t handle, int64_t csr_mode) {
    std::shared_ptr<LayerCall> c;
    {
        std::lock_guard<std::mutex> l(g_mu);
        auto it = g_layer.find(handle);
        TORCH_CHECK(it != g_layer.end(), "cwn::layer_fused: unknown handle ", handle);
        c = it->second;
    }
    return c->run(xs.vec(), (int)csr_mode, std::nullopt);
}

std::vector<at::Tensor> op_update_mlp(at::TensorList xs_up, at::TensorList xs_b, int64_t handle) {
    std::shared_ptr<MlpCall> c;
    {
        std::lock_guard<std::mutex> l(g_mu);
        auto it = g_mlp.find(handle);
        TORCH_CHECK(it != g_mlp.end(), "cwn::update_mlp: unknown handle ", handle);
        c = it->second;
    }
    py::gil_scoped_acquire gil;
    py::object r = c->run(xs_up.vec(), xs_b.vec());
    TORCH_CHECK(!r.is_none(), "cwn::update_mlp: these inputs are not what the prepared launch takes");
    return r.cast<std::vector<at::Tensor>>();
}

}  // namespace

TORCH_LIBRARY(cwn, m) {
    m.def("layer_fused(Tensor[] xs, int handle, int csr_mode) -> Tensor[]");
    m.def("update_mlp(Tensor[] xs_up, Tensor[] xs_b, int handle) -> Tensor[]");
}
TORCH_LIBRARY_IMPL(cwn, CUDA, m) {
    m.impl("layer_fused", &op_layer_fused);
    m.impl("update_mlp", &op_update_mlp);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "compiled binding of the eager path (prepared launches of cwn_layer_fused_f32 / cwn_update_mlp_f32)";
    m.attr("abi_version") = CWN_ABI_VERSION;
    py::class_<TensorMarks>(m, "TensorMarks")
        .def(py::init<std::vector<at::Tensor>>())
        .def("current", &TensorMarks::current)
        .def("__len__", &TensorMarks::size);
    py::class_<LayerCall, std::shared_ptr<LayerCall>>(m, "LayerCall")
        .def(py::init<uintptr_t, int64_t, int, int, std::vector<int64_t>, uintptr_t, uintptr_t, int, int>())
        .def("has_plans", &LayerCall::has_plans)
        .def("set_plans", &LayerCall::set_plans)
        .def("run", &LayerCall::run, py::arg("xs"), py::arg("csr_mode") = 0, py::arg("ys") = py::none());
    py::class_<MlpCall, std::shared_ptr<MlpCall>>(m, "MlpCall")
        .def(py::init<uintptr_t, int64_t, int, int, int64_t, uintptr_t, int, std::vector<at::Tensor>, int64_t, int64_t>())
        .def("current", &MlpCall::current)
        .def("run", &MlpCall::run);
    m.def("layer_register", &layer_register);
    m.def("mlp_register", &mlp_register);
    m.def("release", &release);
}
