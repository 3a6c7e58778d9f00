// torch_glue.cpp -- the autograd plumbing of the in-batch loss as a C++ torch extension (blp_amd/_torch_glue*.so).
//
// No kernel, no HIP call, no arithmetic: this file only moves what blp_amd/ops.py's Python autograd.Function moved -- tensor
// pointers, sizes and the caller's stream -- into blp_inbatch_loss_fwd / _bwd_t of the C-ABI (include/blp_hip.h), which it
// reaches through function pointers handed over by blp_amd._lib (the library ctypes already loaded: one instance, no link
// dependency).  Why it exists: LinkPrediction.compute_loss (models.py:51-70) is called once per training step, its
// kernels take ~20 us, and a Python autograd.Function cost the caller 140 us per step around them (round 3: apply() +
// engine -> Python backward() + five torch.empty + two ctypes calls).  A torch::autograd::Function costs a few us.
//
// Built by blp_amd/build.py with the host compiler against the installed torch; if it is missing (another torch build),
// blp_amd.ops falls back to the Python plumbing -- the kernels are the same either way.
#include <torch/extension.h>

#include <cstdint>
#include <map>
#include <mutex>
#include <utility>

namespace {

using fwd_fn = int (*)(int, int, int, int, const void*, const void*, const int64_t*, int, int, int, float, float*, float*, float*,
                       int32_t*, int, void*);
using size_fn = size_t (*)(int, int, int, int);
using bwd_fn = int (*)(int, int, int, int, const void*, const void*, const int64_t*, int, int, int, float, const float*,
                       const float*, const float*, void*, void*, int, void*);
using err_fn = const char* (*)();

fwd_fn g_fwd = nullptr;
bwd_fn g_bwd = nullptr;
err_fn g_err = nullptr;
size_fn g_save_floats = nullptr;

// The forward's ticket counter (include/blp_hip.h: BLP_INBATCH_TICKET_INTS = 4 int32, zero on entry, left zero by the kernel): one
// tensor per (device, stream), zeroed when first asked for; calls on one stream are ordered and share it.
std::mutex g_ticket_mutex;
std::map<std::pair<int, int64_t>, at::Tensor> g_tickets;

int32_t* ticket_for(const at::Tensor& like, int64_t stream) {
    const std::pair<int, int64_t> key(like.get_device(), stream);
    std::lock_guard<std::mutex> lock(g_ticket_mutex);
    auto it = g_tickets.find(key);
    if (it == g_tickets.end()) it = g_tickets.emplace(key, at::zeros({4}, like.options().dtype(at::kInt))).first;
    return it->second.data_ptr<int32_t>();
}

int dtype_id(at::ScalarType t) {  // BLP_DTYPE_* of include/blp_hip.h
    switch (t) {
        case at::kFloat: return 0;
        case at::kHalf: return 1;
        case at::kBFloat16: return 2;
        default: return -1;
    }
}

constexpr int64_t kAbiVersion = 60000;  // the include/blp_hip.h this file was written against (major checked at bind time by blp_amd.ops)

struct InBatchLoss : public torch::autograd::Function<InBatchLoss> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& ent_embs, const at::Tensor& rel_vecs,
                              const at::Tensor& neg_idx, int64_t model, int64_t loss_id, double regularizer, int64_t stream) {
        TORCH_CHECK(g_fwd && g_bwd, "blp_amd torch glue: bind() was not called");
        TORCH_CHECK(ent_embs.is_cuda() && rel_vecs.is_cuda() && neg_idx.is_cuda(),
                    "blp_amd.ops works on HIP device tensors only; there is no CPU fallback in the product path");
        TORCH_CHECK(ent_embs.dim() == 3 && ent_embs.size(1) == 2, "ent_embs must be (B, 2, D)");
        const int64_t B = ent_embs.size(0), D = ent_embs.size(2);
        TORCH_CHECK(neg_idx.dim() == 3 && neg_idx.size(0) == B && neg_idx.size(2) == 2, "neg_idx must be (B, K, 2)");
        const int64_t K = neg_idx.size(1);
        const int ent_t = dtype_id(ent_embs.scalar_type()), rel_t = dtype_id(rel_vecs.scalar_type());
        TORCH_CHECK(ent_t >= 0 && (rel_t == ent_t || rel_t == 0),
                    "ent_embs must be float32 / float16 / bfloat16 and rel_vecs of the same dtype or float32");
        TORCH_CHECK(rel_vecs.numel() == B * D, "rel_vecs must hold B rows of D elements");
        const at::Tensor ent = ent_embs.contiguous();
        const at::Tensor rel = rel_vecs.reshape({B, D}).contiguous();
        const at::Tensor idx = (neg_idx.scalar_type() == at::kLong ? neg_idx : neg_idx.to(at::kLong)).contiguous();
        const auto f32 = ent.options().dtype(at::kFloat);
        at::Tensor loss = at::empty({}, f32);
        // positives' scores, the workgroups' partial loss sums, the index of neg_idx the backward walks
        at::Tensor pos = at::empty({(int64_t)g_save_floats((int)model, (int)B, (int)K, (int)D)}, f32);
        at::Tensor neg = at::empty({B, K}, f32);
        const int device = ent.get_device();
        const int rc = g_fwd((int)model, (int)loss_id, ent_t, rel_t, ent.data_ptr(), rel.data_ptr(), idx.data_ptr<int64_t>(), (int)B,
                             (int)K, (int)D, (float)regularizer, loss.data_ptr<float>(), pos.data_ptr<float>(),
                             neg.data_ptr<float>(), ticket_for(ent, stream), device, reinterpret_cast<void*>(stream));
        TORCH_CHECK(rc == 0, "blp_inbatch_loss_fwd failed with status ", rc, ": ", g_err ? g_err() : "");
        ctx->save_for_backward({ent, rel, idx, pos, neg});
        ctx->saved_data["model"] = model;
        ctx->saved_data["loss"] = loss_id;
        ctx->saved_data["reg"] = regularizer;
        ctx->saved_data["stream"] = stream;
        ctx->saved_data["rel_shape"] = rel_vecs.sizes().vec();
        return loss;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &ent = saved[0], &rel = saved[1], &idx = saved[2], &pos = saved[3], &neg = saved[4];
        const int64_t B = ent.size(0), D = ent.size(2), K = idx.size(1);
        at::Tensor g = grads[0];
        if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
        g = g.contiguous();
        at::Tensor grad_ent = at::empty_like(ent);
        at::Tensor grad_rel = at::empty_like(rel);
        // (the engine runs a backward node on the stream its forward ran on: the stream captured there is the current one)
        const int64_t stream = ctx->saved_data["stream"].toInt();
        const int rc = g_bwd((int)ctx->saved_data["model"].toInt(), (int)ctx->saved_data["loss"].toInt(), dtype_id(ent.scalar_type()),
                             dtype_id(rel.scalar_type()), ent.data_ptr(), rel.data_ptr(), idx.data_ptr<int64_t>(), (int)B, (int)K, (int)D,
                             (float)ctx->saved_data["reg"].toDouble(), g.data_ptr<float>(), pos.data_ptr<float>(),
                             neg.data_ptr<float>(), grad_ent.data_ptr(), grad_rel.data_ptr(), ent.get_device(),
                             reinterpret_cast<void*>(stream));
        TORCH_CHECK(rc == 0, "blp_inbatch_loss_bwd failed with status ", rc, ": ", g_err ? g_err() : "");
        return {grad_ent, grad_rel.reshape(ctx->saved_data["rel_shape"].toIntVector()), at::Tensor(), at::Tensor(), at::Tensor(),
                at::Tensor(), at::Tensor()};
    }
};

// What the autograd machinery itself costs around a fused loss of this shape: the same allocations, the same saved tensors, the
// same engine round trip -- and no kernel at all.  bench.py times it next to the real function; the difference is what the
// C-ABI calls (argument checks + three launches) add on the host.
struct AutogradFloor : public torch::autograd::Function<AutogradFloor> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& ent_embs, const at::Tensor& rel_vecs,
                              const at::Tensor& neg_idx) {
        const int64_t B = ent_embs.size(0), K = neg_idx.size(1);
        const auto f32 = ent_embs.options().dtype(at::kFloat);
        at::Tensor loss = at::empty({}, f32), pos = at::empty({(int64_t)g_save_floats(0, (int)B, (int)K, (int)ent_embs.size(2))}, f32),
                   neg = at::empty({B, K}, f32);
        ctx->save_for_backward({ent_embs, rel_vecs, neg_idx, pos, neg});
        return loss;
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        return {at::empty_like(saved[0]), at::empty_like(saved[1]), at::Tensor()};
    }
};

// (tools/autograd_floor_probe.py: which part of such a node costs what.  flags: 1 = save nothing, 2 = no scratch tensors in forward,
//  4 = no gradient for rel_vecs, 8 = none for ent_embs, 16 = gradients are plain at::empty of the saved sizes, not empty_like)
struct AutogradFloorV : public torch::autograd::Function<AutogradFloorV> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& ent_embs, const at::Tensor& rel_vecs,
                              const at::Tensor& neg_idx, int64_t flags) {
        const int64_t B = ent_embs.size(0), K = neg_idx.size(1);
        const auto f32 = ent_embs.options().dtype(at::kFloat);
        at::Tensor loss = at::empty({}, f32);
        if (!(flags & 2)) {
            at::Tensor pos = at::empty({(int64_t)g_save_floats(0, (int)B, (int)K, (int)ent_embs.size(2))}, f32), neg = at::empty({B, K}, f32);
            if (!(flags & 1)) ctx->save_for_backward({ent_embs, rel_vecs, neg_idx, pos, neg});
        } else if (!(flags & 1)) {
            ctx->save_for_backward({ent_embs, rel_vecs, neg_idx});
        }
        ctx->saved_data["flags"] = flags;
        ctx->saved_data["e_shape"] = ent_embs.sizes().vec();
        ctx->saved_data["r_shape"] = rel_vecs.sizes().vec();
        ctx->saved_data["dev"] = (int64_t)ent_embs.get_device();
        return loss;
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const int64_t flags = ctx->saved_data["flags"].toInt();
        const auto opt = at::TensorOptions().dtype(at::kFloat).device(at::kCUDA, (c10::DeviceIndex)ctx->saved_data["dev"].toInt());
        at::Tensor ge, gr;
        if (flags & 16 || flags & 1) {
            if (!(flags & 8)) ge = at::empty(ctx->saved_data["e_shape"].toIntVector(), opt);
            if (!(flags & 4)) gr = at::empty(ctx->saved_data["r_shape"].toIntVector(), opt);
        } else {
            const auto saved = ctx->get_saved_variables();
            if (!(flags & 8)) ge = at::empty_like(saved[0]);
            if (!(flags & 4)) gr = at::empty_like(saved[1]);
        }
        return {ge, gr, at::Tensor(), at::Tensor()};
    }
};

at::Tensor autograd_floor_variant(const at::Tensor& ent_embs, const at::Tensor& rel_vecs, const at::Tensor& neg_idx, int64_t flags) {
    return AutogradFloorV::apply(ent_embs, rel_vecs, neg_idx, flags);
}

at::Tensor autograd_floor(const at::Tensor& ent_embs, const at::Tensor& rel_vecs, const at::Tensor& neg_idx) {
    return AutogradFloor::apply(ent_embs, rel_vecs, neg_idx);
}

void bind(uintptr_t fwd, uintptr_t bwd, uintptr_t err, uintptr_t save_floats) {
    g_fwd = reinterpret_cast<fwd_fn>(fwd);
    g_bwd = reinterpret_cast<bwd_fn>(bwd);
    g_err = reinterpret_cast<err_fn>(err);
    g_save_floats = reinterpret_cast<size_fn>(save_floats);
}

at::Tensor inbatch_loss(const at::Tensor& ent_embs, const at::Tensor& rel_vecs, const at::Tensor& neg_idx, int64_t model,
                        int64_t loss_id, double regularizer, int64_t stream) {
    return InBatchLoss::apply(ent_embs, rel_vecs, neg_idx, model, loss_id, regularizer, stream);
}

}  // namespace

PYBIND11_MODULE(_torch_glue, m) {
    m.doc() = "blp_amd: C++ autograd plumbing around the C-ABI's in-batch loss (no kernels here)";
    m.def("bind", &bind, "hand over the addresses of blp_inbatch_loss_fwd, blp_inbatch_loss_bwd, blp_last_error, blp_inbatch_loss_save_floats");
    m.def("inbatch_loss", &inbatch_loss, py::call_guard<py::gil_scoped_release>(),
          "compute_loss on in-batch negatives: (ent_embs, rel_vecs, neg_idx, model_id, loss_id, regularizer, raw_stream) -> loss");
    m.def("autograd_floor", &autograd_floor, py::call_guard<py::gil_scoped_release>(),
          "a node of the same shape that launches nothing (bench.py: what autograd itself costs per step)");
    m.def("autograd_floor_variant", &autograd_floor_variant, py::call_guard<py::gil_scoped_release>(), "autograd_floor with parts switched off (probe)");
    m.attr("abi_version") = kAbiVersion;
}
