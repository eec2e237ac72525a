// Compiled host binding of the module path (news_recsys_amd/ops.py): one call validates a batch's tensors, refreshes the
// nrx_feature_t descriptors, allocates the outputs and enqueues nrx_embed_fwd_train -- what ops._FastForward / _EmbedFn.forward
// do through ~150 Python attribute reads and a ctypes call per batch (22 us for the 26-feature C2 plan; 61 us with the
// autograd bookkeeping around it: profiles/r02_host_overhead.txt).  The reference surface being served:
// BaseModel.get_embeddings_from_batch (src/model/BaseModel/base_model.py:284-308).
//
// Built with g++ against libtorch (ATen is used for tensor METADATA and output allocation only -- no device code here, the
// kernels stay behind the C-ABI of libnrx_hip.so, whose entry point arrives as an address).  Optional: without this module
// ops.py keeps its ctypes path (host marshalling only; there is no compute fallback anywhere).
#include <torch/extension.h>
#include <torch/csrc/autograd/python_variable.h>

#include <vector>

#include "nrx_embed.h"

namespace py = pybind11;

namespace {

using fwd_train_fn = int (*)(const nrx_feature_t*, int32_t, int64_t, float*, int64_t, float*, int64_t, float*, float*, int64_t,
                             int32_t*, void*);

struct BoundPlan {
    std::vector<nrx_feature_t> f;
    std::vector<int> table_of;
    int64_t out_width = 0, wide_width = 0;
    bool use_fm = false;
    int fm_dim = 0;
    fwd_train_fn fwd = nullptr;
    // what the descriptors were last bound to: every table's data pointer and row count (compared on every call -- a list's
    // address says nothing, CPython reuses it, and a table in the MIDDLE of the list may be re-allocated or resized)
    std::vector<const void*> bound_ptr;
    std::vector<int64_t> bound_rows;
    c10::Device device{c10::kCPU};

    // slots: sequence of (kind, table, dim, bag_len, out_col, wide_col, fm_field, flags)
    BoundPlan(const py::sequence& slots, int64_t out_w, int64_t wide_w, bool fm, uint64_t fwd_addr)
        : out_width(out_w), wide_width(wide_w), use_fm(fm), fwd(reinterpret_cast<fwd_train_fn>(fwd_addr)) {
        for (const auto& h : slots) {
            const auto s = h.cast<py::tuple>();
            nrx_feature_t d;
            memset(&d, 0, sizeof(d));
            d.kind = s[0].cast<int>();
            table_of.push_back(s[1].cast<int>());
            d.dim = s[2].cast<int>();
            d.bag_len = s[3].cast<int>();
            d.out_col = s[4].cast<int>();
            d.wide_col = s[5].cast<int>();
            d.fm_field = s[6].cast<int>();
            d.flags = s[7].cast<int>();
            if (d.fm_field && d.dim > fm_dim) fm_dim = d.dim;
            f.push_back(d);
        }
    }

    bool bind_tables(PyObject* tables) {
        const Py_ssize_t nt = PyList_GET_SIZE(tables);
        std::vector<const at::Tensor*> ts((size_t)nt);
        for (Py_ssize_t i = 0; i < nt; ++i) {
            PyObject* o = PyList_GET_ITEM(tables, i);
            if (!THPVariable_Check(o)) return false;
            const at::Tensor& t = THPVariable_Unpack(o);
            if (!t.is_cuda() || t.scalar_type() != at::kFloat || t.dim() != 2 || !t.is_contiguous()) return false;
            ts[(size_t)i] = &t;
        }
        for (size_t i = 0; i < f.size(); ++i) {
            const int k = table_of[i];
            if (f[i].kind == NRX_DENSE) { f[i].table = nullptr; f[i].rows = 0; continue; }
            if (k < 0 || k >= nt) return false;
            f[i].table = static_cast<const float*>(ts[(size_t)k]->data_ptr());
            f[i].rows = ts[(size_t)k]->size(0);
        }
        bound_ptr.resize((size_t)nt);
        bound_rows.resize((size_t)nt);
        for (Py_ssize_t i = 0; i < nt; ++i) {
            bound_ptr[(size_t)i] = ts[(size_t)i]->data_ptr();
            bound_rows[(size_t)i] = ts[(size_t)i]->size(0);
        }
        if (nt) device = ts[0]->device();
        return true;
    }

    // True when every table of the list is the storage (and row count) the descriptors hold.
    bool tables_unchanged(PyObject* tables) const {
        const Py_ssize_t nt = PyList_GET_SIZE(tables);
        if ((size_t)nt != bound_ptr.size()) return false;
        for (Py_ssize_t i = 0; i < nt; ++i) {
            PyObject* o = PyList_GET_ITEM(tables, i);
            if (!THPVariable_Check(o)) return false;
            const at::Tensor& t = THPVariable_Unpack(o);
            if (t.dim() != 2 || t.data_ptr() != bound_ptr[(size_t)i] || t.size(0) != bound_rows[(size_t)i]) return false;
        }
        return true;
    }

    // Returns None when the batch needs a conversion the Python path knows how to make (odd dtypes, non-contiguous tensors,
    // CSR bags, ...); else (B, out | None, wide | None, fm | None, fm_sums | None).
    py::object forward(py::handle tables, py::handle inputs, py::handle weights, int64_t out_ld, bool need_out, uint64_t status_ptr,
                       uint64_t stream, bool want_sums) {
        if (!PyList_Check(tables.ptr()) || !PyList_Check(inputs.ptr()) || !PyList_Check(weights.ptr())) return py::none();
        PyObject* tl = tables.ptr();
        const size_t n = f.size();
        if ((size_t)PyList_GET_SIZE(inputs.ptr()) != n || (size_t)PyList_GET_SIZE(weights.ptr()) != n || n == 0) return py::none();
        if (!tables_unchanged(tl) && !bind_tables(tl)) return py::none();
        int64_t B = -1;
        for (size_t i = 0; i < n; ++i) {
            PyObject* xo = PyList_GET_ITEM(inputs.ptr(), (Py_ssize_t)i);
            if (!THPVariable_Check(xo)) return py::none();
            const at::Tensor& x = THPVariable_Unpack(xo);
            nrx_feature_t& d = f[i];
            if (!x.is_cuda() || !x.is_contiguous() || (d.flags & NRX_FEAT_BAG_CSR)) return py::none();
            const auto st = x.scalar_type();
            const int64_t b = x.dim() >= 1 ? x.size(0) : -1;
            if (B < 0) B = b;
            if (b != B) return py::none();
            PyObject* wo = PyList_GET_ITEM(weights.ptr(), (Py_ssize_t)i);
            d.weight = nullptr;
            if (d.kind == NRX_DENSE) {
                if (x.dim() != 1 || (st != at::kFloat && st != at::kDouble)) return py::none();
                d.index_bits = st == at::kDouble ? 64 : 32;
            } else {
                if (st == at::kLong) d.index_bits = 64;
                else if (st == at::kInt) d.index_bits = 32;
                else return py::none();
                if (d.kind == NRX_SPARSE) {
                    if (x.dim() != 1) return py::none();
                } else {
                    if (x.dim() != 2 || x.size(1) != d.bag_len) return py::none();
                    if (wo != Py_None) {
                        if (!THPVariable_Check(wo)) return py::none();
                        const at::Tensor& w = THPVariable_Unpack(wo);
                        if (!w.is_cuda() || !w.is_contiguous() || w.scalar_type() != at::kFloat || w.dim() != 2 || w.size(0) != B ||
                            w.size(1) != d.bag_len)
                            return py::none();
                        d.weight = static_cast<const float*>(w.data_ptr());
                    } else if (d.kind != NRX_BAG_MEAN) {
                        return py::none();          // masked mean / weighted sum without weights: the Python path raises the right error
                    }
                }
            }
            d.index = x.data_ptr();
        }
        if (B < 0) return py::none();
        const int64_t ld = out_ld > 0 ? out_ld : out_width;
        if (ld < out_width) throw py::value_error("out_ld smaller than the plan's out_width");
        const auto opt = at::TensorOptions().dtype(at::kFloat).device(device);
        at::Tensor out, wide, fm, sums;
        if (need_out) out = at::empty({B, ld}, opt);
        if (wide_width) wide = at::empty({B, wide_width}, opt);
        if (use_fm) fm = at::empty({B}, opt);
        if (want_sums && use_fm && need_out && fm_dim) sums = at::empty({B, (int64_t)fm_dim}, opt);
        if (B > 0) {
            const int rc = fwd(f.data(), (int32_t)n, B, need_out ? out.data_ptr<float>() : nullptr, ld,
                               wide_width ? wide.data_ptr<float>() : nullptr, wide_width, use_fm ? fm.data_ptr<float>() : nullptr,
                               sums.defined() ? sums.data_ptr<float>() : nullptr, fm_dim, reinterpret_cast<int32_t*>(status_ptr),
                               reinterpret_cast<void*>(stream));
            if (rc != 0) return py::int_(rc);        // the caller turns the status code into the library's exception
        }
        auto wrap = [](const at::Tensor& t) -> py::object {
            return t.defined() ? py::reinterpret_steal<py::object>(THPVariable_Wrap(t)) : py::none();
        };
        return py::make_tuple(B, wrap(out), wrap(wide), wrap(fm), wrap(sums));
    }
};

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "compiled host binding of news_recsys_amd's module path (descriptor packing + launch of nrx_embed_fwd_train)";
    py::class_<BoundPlan>(m, "BoundPlan")
        .def(py::init<const py::sequence&, int64_t, int64_t, bool, uint64_t>())
        .def("forward", &BoundPlan::forward);
}
