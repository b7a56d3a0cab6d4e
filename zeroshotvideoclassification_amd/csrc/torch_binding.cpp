// torch_binding.cpp -- the thin torch-extension layer BASELINE.json's north_star names ("bound through a thin torch
// cpp_extension C-ABI layer"): `torch.ops.zsv.*` operators over the C ABI of libzsv_hip.so (include/zsv_hip.h).
//
// Host C++ only (built with g++ against the torch headers; no device code, nothing generated): every operator builds the POD
// descriptor from the tensor sizes, takes the stream torch is currently recording on, allocates the workspace from torch's caching
// allocator and calls the same `zsv_*` entry point the ctypes glue (`_lib.py` / `ops.py`) calls.  The differentiable forms
// (`zsv::conv3d`, `zsv::batch_norm_relu`, `zsv::relu`, `zsv::linear`) register an Autograd kernel, so C++ / TorchScript callers get the reference's
// `nn.Conv3d` (resnet.py:23-30,40-52; network.py:102-117) and `nn.BatchNorm3d (+ ReLU)` (resnet.py:46-49,94-98) semantics
// without Python.  The training harness of this repo keeps the ctypes path (panel cache, side-stream weight gradients, fused
// block tails live in ops.py); tests/test_torch_binding_gpu.py checks both bindings against each other bit for bit.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/autograd.h>
#include <torch/library.h>

#include <vector>

#include "zsv_hip.h"

namespace {

void ok(int status, const char* what) { TORCH_CHECK(status == 0, what, ": ", zsv_status_string(status)); }

void* stream_of(const at::Tensor& t) {
    return static_cast<void*>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

void want(const at::Tensor& t, const char* name, int64_t dim) {
    TORCH_CHECK(t.is_cuda(), name, " must live on the GPU");       // (ROCm devices are `cuda` devices to torch)
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32");
    TORCH_CHECK(t.dim() == dim, name, " must have ", dim, " dimensions");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous (NCDHW)");
}

at::Tensor scratch(size_t bytes, const at::Tensor& like) {
    return at::empty({static_cast<int64_t>(bytes ? bytes : 1)}, like.options().dtype(at::kByte));
}

zsv_conv_desc describe(at::IntArrayRef x, at::IntArrayRef w, at::IntArrayRef stride, at::IntArrayRef padding) {
    TORCH_CHECK(x.size() == 5 && w.size() == 5 && stride.size() == 3 && padding.size() == 3,
                "conv3d: x and w are 5-d, stride and padding have three entries");
    TORCH_CHECK(x[1] == w[1], "conv3d: x has ", x[1], " channels, w expects ", w[1]);
    zsv_conv_desc d;
    d.N = (int32_t)x[0]; d.Cin = (int32_t)x[1]; d.Ti = (int32_t)x[2]; d.Hi = (int32_t)x[3]; d.Wi = (int32_t)x[4];
    d.Cout = (int32_t)w[0]; d.kT = (int32_t)w[2]; d.kH = (int32_t)w[3]; d.kW = (int32_t)w[4];
    d.sT = (int32_t)stride[0]; d.sH = (int32_t)stride[1]; d.sW = (int32_t)stride[2];
    d.pT = (int32_t)padding[0]; d.pH = (int32_t)padding[1]; d.pW = (int32_t)padding[2];
    d.To = (d.Ti + 2 * d.pT - d.kT) / d.sT + 1;
    d.Ho = (d.Hi + 2 * d.pH - d.kH) / d.sH + 1;
    d.Wo = (d.Wi + 2 * d.pW - d.kW) / d.sW + 1;
    TORCH_CHECK(d.To > 0 && d.Ho > 0 && d.Wo > 0, "conv3d: empty output");
    return d;
}

// ---- plain operators: one C-ABI call each -----------------------------------------------------------------------------------
at::Tensor conv3d_fwd(const at::Tensor& x, const at::Tensor& w, const c10::optional<at::Tensor>& bias, at::IntArrayRef stride,
                      at::IntArrayRef padding, bool relu) {
    want(x, "x", 5);
    want(w, "w", 5);
    const zsv_conv_desc d = describe(x.sizes(), w.sizes(), stride, padding);
    const float* b = nullptr;
    if (bias.has_value() && bias->defined()) {
        want(*bias, "bias", 1);
        TORCH_CHECK(bias->numel() == d.Cout, "conv3d: bias has ", bias->numel(), " entries for ", d.Cout, " channels");
        b = bias->data_ptr<float>();
    }
    at::Tensor y = at::empty({d.N, d.Cout, d.To, d.Ho, d.Wo}, x.options());
    const size_t bytes = zsv_conv3d_fwd_workspace_bytes(&d);
    at::Tensor ws = scratch(bytes, x);
    ok(zsv_conv3d_fwd(&d, x.data_ptr<float>(), w.data_ptr<float>(), b, y.data_ptr<float>(), relu ? 1 : 0, ws.data_ptr(), bytes,
                      stream_of(x)),
       "zsv_conv3d_fwd");
    return y;
}

at::Tensor conv3d_dgrad(const at::Tensor& dy, const at::Tensor& w, at::IntArrayRef x_sizes, at::IntArrayRef stride,
                        at::IntArrayRef padding) {
    want(dy, "dy", 5);
    want(w, "w", 5);
    const zsv_conv_desc d = describe(x_sizes, w.sizes(), stride, padding);
    TORCH_CHECK(dy.size(0) == d.N && dy.size(1) == d.Cout && dy.size(2) == d.To && dy.size(3) == d.Ho && dy.size(4) == d.Wo,
                "conv3d_dgrad: dy does not have the output's shape");
    at::Tensor dx = at::empty(x_sizes, dy.options());
    const size_t bytes = zsv_conv3d_dgrad_workspace_bytes(&d);
    at::Tensor ws = scratch(bytes, dy);
    ok(zsv_conv3d_dgrad(&d, dy.data_ptr<float>(), w.data_ptr<float>(), dx.data_ptr<float>(), ws.data_ptr(), bytes, stream_of(dy)),
       "zsv_conv3d_dgrad");
    return dx;
}

at::Tensor conv3d_wgrad(const at::Tensor& x, const at::Tensor& dy, at::IntArrayRef w_sizes, at::IntArrayRef stride,
                        at::IntArrayRef padding) {
    want(x, "x", 5);
    want(dy, "dy", 5);
    const zsv_conv_desc d = describe(x.sizes(), w_sizes, stride, padding);
    TORCH_CHECK(dy.size(0) == d.N && dy.size(1) == d.Cout && dy.size(2) == d.To && dy.size(3) == d.Ho && dy.size(4) == d.Wo,
                "conv3d_wgrad: dy does not have the output's shape");
    at::Tensor dw = at::empty(w_sizes, x.options());
    const size_t bytes = zsv_conv3d_wgrad_workspace_bytes(&d);
    at::Tensor ws = scratch(bytes, x);
    ok(zsv_conv3d_wgrad(&d, x.data_ptr<float>(), dy.data_ptr<float>(), dw.data_ptr<float>(), ws.data_ptr(), bytes, stream_of(x)),
       "zsv_conv3d_wgrad");
    return dw;
}

// training-mode BatchNorm3d (+ ReLU): y, save_mean, save_invstd; running statistics updated in place like torch
std::tuple<at::Tensor, at::Tensor, at::Tensor> bn_train_fwd(const at::Tensor& x, const at::Tensor& gamma, const at::Tensor& beta,
                                                            at::Tensor running_mean, at::Tensor running_var, double momentum,
                                                            double eps, bool relu) {
    want(x, "x", 5);
    want(gamma, "weight", 1);
    want(beta, "bias", 1);
    want(running_mean, "running_mean", 1);
    want(running_var, "running_var", 1);
    const int32_t N = (int32_t)x.size(0), C = (int32_t)x.size(1), S = (int32_t)(x.size(2) * x.size(3) * x.size(4));
    TORCH_CHECK(gamma.numel() == C && beta.numel() == C && running_mean.numel() == C && running_var.numel() == C,
                "batch_norm: per-channel tensors must have ", C, " entries");
    at::Tensor y = at::empty_like(x), mean = at::empty({C}, x.options()), invstd = at::empty({C}, x.options());
    const size_t bytes = zsv_bn_workspace_bytes(N, C, S);
    at::Tensor ws = scratch(bytes, x);
    ok(zsv_bn_fwd_train(x.data_ptr<float>(), N, C, S, gamma.data_ptr<float>(), beta.data_ptr<float>(), nullptr, relu ? 1 : 0,
                        y.data_ptr<float>(), mean.data_ptr<float>(), invstd.data_ptr<float>(), running_mean.data_ptr<float>(),
                        running_var.data_ptr<float>(), (float)momentum, (float)eps, ws.data_ptr(), bytes, stream_of(x)),
       "zsv_bn_fwd_train");
    return {y, mean, invstd};
}

// backward of the op above: dx, dgamma, dbeta (ReLU mask from the saved output)
std::tuple<at::Tensor, at::Tensor, at::Tensor> bn_train_bwd(const at::Tensor& dy, const at::Tensor& x, const at::Tensor& y,
                                                            const at::Tensor& gamma, const at::Tensor& beta, const at::Tensor& mean,
                                                            const at::Tensor& invstd, bool relu) {
    want(dy, "dy", 5);
    want(x, "x", 5);
    want(y, "y", 5);
    const int32_t N = (int32_t)x.size(0), C = (int32_t)x.size(1), S = (int32_t)(x.size(2) * x.size(3) * x.size(4));
    at::Tensor dx = at::empty_like(x), dgamma = at::empty({C}, x.options()), dbeta = at::empty({C}, x.options());
    const size_t bytes = zsv_bn_workspace_bytes(N, C, S);
    at::Tensor ws = scratch(bytes, x);
    ok(zsv_bn_bwd(dy.data_ptr<float>(), x.data_ptr<float>(), y.data_ptr<float>(), N, C, S, gamma.data_ptr<float>(),
                  beta.data_ptr<float>(), mean.data_ptr<float>(), invstd.data_ptr<float>(), relu ? 1 : 0, dx.data_ptr<float>(),
                  nullptr, dgamma.data_ptr<float>(), dbeta.data_ptr<float>(), ws.data_ptr(), bytes, stream_of(x)),
       "zsv_bn_bwd");
    return {dx, dgamma, dbeta};
}

// nn.ReLU (resnet.py:49; network.py:147-166,614) and its backward (mask from the saved output, like ReLU(inplace=True))
at::Tensor relu_fwd(const at::Tensor& x) {
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.is_contiguous(), "relu: contiguous float32 GPU tensor expected");
    at::Tensor y = at::empty_like(x);
    ok(zsv_relu_fwd(x.data_ptr<float>(), y.data_ptr<float>(), x.numel(), stream_of(x)), "zsv_relu_fwd");
    return y;
}

at::Tensor relu_bwd(const at::Tensor& dy, const at::Tensor& y) {
    TORCH_CHECK(dy.is_cuda() && y.is_cuda() && dy.scalar_type() == at::kFloat && y.scalar_type() == at::kFloat && dy.is_contiguous() &&
                    y.is_contiguous() && dy.numel() == y.numel(), "relu_bwd: two contiguous float32 GPU tensors of one size expected");
    at::Tensor dx = at::empty_like(dy);
    ok(zsv_relu_bwd(dy.data_ptr<float>(), y.data_ptr<float>(), dx.data_ptr<float>(), dy.numel(), stream_of(dy)), "zsv_relu_bwd");
    return dx;
}

// nn.Linear (+ ReLU) of network.MLP (network.py:603-617): y = relu?(x w^T + b), x (rows, in), w (out, in)
at::Tensor linear_fwd(const at::Tensor& x, const at::Tensor& w, const c10::optional<at::Tensor>& bias, bool relu) {
    want(x, "x", 2);
    want(w, "w", 2);
    TORCH_CHECK(x.size(1) == w.size(1), "linear: x has ", x.size(1), " features, w expects ", w.size(1));
    const int32_t rows = (int32_t)x.size(0), in = (int32_t)x.size(1), out = (int32_t)w.size(0);
    const float* b = nullptr;
    if (bias.has_value() && bias->defined()) {
        want(*bias, "bias", 1);
        TORCH_CHECK(bias->numel() == out, "linear: bias has ", bias->numel(), " entries for ", out, " outputs");
        b = bias->data_ptr<float>();
    }
    at::Tensor y = at::empty({rows, out}, x.options());
    const size_t bytes = zsv_linear_fwd_workspace_bytes(rows, in, out);
    at::Tensor ws = scratch(bytes, x);
    ok(zsv_linear_fwd(x.data_ptr<float>(), w.data_ptr<float>(), b, y.data_ptr<float>(), rows, in, out, relu ? 1 : 0, ws.data_ptr(), bytes,
                      stream_of(x)),
       "zsv_linear_fwd");
    return y;
}

at::Tensor linear_dgrad(const at::Tensor& dy, const at::Tensor& w) {
    want(dy, "dy", 2);
    want(w, "w", 2);
    TORCH_CHECK(dy.size(1) == w.size(0), "linear_dgrad: dy has ", dy.size(1), " columns, w has ", w.size(0), " rows");
    const int32_t rows = (int32_t)dy.size(0), in = (int32_t)w.size(1), out = (int32_t)w.size(0);
    at::Tensor dx = at::empty({rows, in}, dy.options());
    const size_t bytes = zsv_linear_dgrad_workspace_bytes(rows, in, out);
    at::Tensor ws = scratch(bytes, dy);
    ok(zsv_linear_dgrad(dy.data_ptr<float>(), w.data_ptr<float>(), dx.data_ptr<float>(), rows, in, out, ws.data_ptr(), bytes, stream_of(dy)),
       "zsv_linear_dgrad");
    return dx;
}

at::Tensor linear_wgrad(const at::Tensor& x, const at::Tensor& dy) {
    want(x, "x", 2);
    want(dy, "dy", 2);
    TORCH_CHECK(x.size(0) == dy.size(0), "linear_wgrad: x and dy must have the same number of rows");
    const int32_t rows = (int32_t)x.size(0), in = (int32_t)x.size(1), out = (int32_t)dy.size(1);
    at::Tensor dw = at::empty({out, in}, x.options());
    const size_t bytes = zsv_linear_wgrad_workspace_bytes(rows, in, out);
    at::Tensor ws = scratch(bytes, x);
    ok(zsv_linear_wgrad(x.data_ptr<float>(), dy.data_ptr<float>(), dw.data_ptr<float>(), rows, in, out, ws.data_ptr(), bytes, stream_of(x)),
       "zsv_linear_wgrad");
    return dw;
}

std::string version() { return zsv_version(); }

// ---- differentiable forms ---------------------------------------------------------------------------------------------------
struct Conv3dFn : public torch::autograd::Function<Conv3dFn> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x, const at::Tensor& w,
                              const c10::optional<at::Tensor>& bias, std::vector<int64_t> stride, std::vector<int64_t> padding) {
        at::AutoDispatchBelowADInplaceOrView guard;
        ctx->save_for_backward({x, w});
        ctx->saved_data["stride"] = stride;
        ctx->saved_data["padding"] = padding;
        ctx->saved_data["has_bias"] = bias.has_value() && bias->defined();
        return conv3d_fwd(x, w, bias, stride, padding, false);
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &x = saved[0], &w = saved[1];
        const auto stride = ctx->saved_data["stride"].toIntVector(), padding = ctx->saved_data["padding"].toIntVector();
        const at::Tensor dy = grads[0].contiguous();
        at::Tensor dx, dw, db;
        if (ctx->needs_input_grad(0)) dx = conv3d_dgrad(dy, w, x.sizes(), stride, padding);
        if (ctx->needs_input_grad(1)) dw = conv3d_wgrad(x, dy, w.sizes(), stride, padding);
        if (ctx->saved_data["has_bias"].toBool() && ctx->needs_input_grad(2)) db = dy.sum({0, 2, 3, 4});
        return {dx, dw, db, at::Tensor(), at::Tensor()};
    }
};

at::Tensor conv3d_autograd(const at::Tensor& x, const at::Tensor& w, const c10::optional<at::Tensor>& bias, at::IntArrayRef stride,
                           at::IntArrayRef padding) {
    return Conv3dFn::apply(x, w, bias, stride.vec(), padding.vec());
}

struct BatchNormReluFn : public torch::autograd::Function<BatchNormReluFn> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x, const at::Tensor& gamma, const at::Tensor& beta,
                              at::Tensor running_mean, at::Tensor running_var, double momentum, double eps, bool relu) {
        at::AutoDispatchBelowADInplaceOrView guard;
        auto [y, mean, invstd] = bn_train_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, relu);
        ctx->save_for_backward({x, y, gamma, beta, mean, invstd});
        ctx->saved_data["relu"] = relu;
        return y;
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto s = ctx->get_saved_variables();
        auto [dx, dgamma, dbeta] = bn_train_bwd(grads[0].contiguous(), s[0], s[1], s[2], s[3], s[4], s[5], ctx->saved_data["relu"].toBool());
        return {dx, dgamma, dbeta, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor batch_norm_relu_autograd(const at::Tensor& x, const at::Tensor& gamma, const at::Tensor& beta, at::Tensor running_mean,
                                    at::Tensor running_var, double momentum, double eps, bool relu) {
    return BatchNormReluFn::apply(x, gamma, beta, running_mean, running_var, momentum, eps, relu);
}

struct ReluFn : public torch::autograd::Function<ReluFn> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x) {
        at::AutoDispatchBelowADInplaceOrView guard;
        at::Tensor y = relu_fwd(x.contiguous());
        ctx->save_for_backward({y});
        return y;
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        return {relu_bwd(grads[0].contiguous(), ctx->get_saved_variables()[0])};
    }
};
at::Tensor relu_autograd(const at::Tensor& x) { return ReluFn::apply(x); }

struct LinearFn : public torch::autograd::Function<LinearFn> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x, const at::Tensor& w,
                              const c10::optional<at::Tensor>& bias) {
        at::AutoDispatchBelowADInplaceOrView guard;
        ctx->save_for_backward({x, w});
        ctx->saved_data["has_bias"] = bias.has_value() && bias->defined();
        return linear_fwd(x, w, bias, false);
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor dy = grads[0].contiguous();
        at::Tensor dx, dw, db;
        if (ctx->needs_input_grad(0)) dx = linear_dgrad(dy, saved[1]);
        if (ctx->needs_input_grad(1)) dw = linear_wgrad(saved[0], dy);
        if (ctx->saved_data["has_bias"].toBool() && ctx->needs_input_grad(2)) db = dy.sum(0);
        return {dx, dw, db};
    }
};
at::Tensor linear_autograd(const at::Tensor& x, const at::Tensor& w, const c10::optional<at::Tensor>& bias) {
    return LinearFn::apply(x, w, bias);
}
at::Tensor linear_plain(const at::Tensor& x, const at::Tensor& w, const c10::optional<at::Tensor>& bias) {
    return linear_fwd(x, w, bias, false);
}

// without autograd in the key set (inference mode): the forward alone
at::Tensor conv3d_plain(const at::Tensor& x, const at::Tensor& w, const c10::optional<at::Tensor>& bias, at::IntArrayRef stride,
                        at::IntArrayRef padding) {
    return conv3d_fwd(x, w, bias, stride, padding, false);
}

at::Tensor batch_norm_relu_plain(const at::Tensor& x, const at::Tensor& gamma, const at::Tensor& beta, at::Tensor running_mean,
                                 at::Tensor running_var, double momentum, double eps, bool relu) {
    return std::get<0>(bn_train_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, relu));
}

}  // namespace

TORCH_LIBRARY(zsv, m) {
    m.def("version() -> str", version);
    m.def("conv3d_fwd(Tensor x, Tensor w, Tensor? bias, int[3] stride, int[3] padding, bool relu=False) -> Tensor");
    m.def("conv3d_dgrad(Tensor dy, Tensor w, int[] x_sizes, int[3] stride, int[3] padding) -> Tensor");
    m.def("conv3d_wgrad(Tensor x, Tensor dy, int[] w_sizes, int[3] stride, int[3] padding) -> Tensor");
    m.def("bn_train_fwd(Tensor x, Tensor weight, Tensor bias, Tensor(a!) running_mean, Tensor(b!) running_var, float momentum, "
          "float eps, bool relu=False) -> (Tensor, Tensor, Tensor)");
    m.def("bn_train_bwd(Tensor dy, Tensor x, Tensor y, Tensor weight, Tensor bias, Tensor save_mean, Tensor save_invstd, "
          "bool relu=False) -> (Tensor, Tensor, Tensor)");
    m.def("relu_fwd(Tensor x) -> Tensor");
    m.def("relu_bwd(Tensor dy, Tensor y) -> Tensor");
    m.def("linear_fwd(Tensor x, Tensor w, Tensor? bias, bool relu=False) -> Tensor");
    m.def("linear_dgrad(Tensor dy, Tensor w) -> Tensor");
    m.def("linear_wgrad(Tensor x, Tensor dy) -> Tensor");
    m.def("relu(Tensor x) -> Tensor");
    m.def("linear(Tensor x, Tensor w, Tensor? bias) -> Tensor");
    m.def("conv3d(Tensor x, Tensor w, Tensor? bias, int[3] stride, int[3] padding) -> Tensor");
    m.def("batch_norm_relu(Tensor x, Tensor weight, Tensor bias, Tensor(a!) running_mean, Tensor(b!) running_var, float momentum, "
          "float eps, bool relu=False) -> Tensor");
}

TORCH_LIBRARY_IMPL(zsv, CUDA, m) {      // torch's name of the HIP backend on ROCm builds
    m.impl("conv3d_fwd", conv3d_fwd);
    m.impl("conv3d_dgrad", conv3d_dgrad);
    m.impl("conv3d_wgrad", conv3d_wgrad);
    m.impl("bn_train_fwd", bn_train_fwd);
    m.impl("bn_train_bwd", bn_train_bwd);
    m.impl("conv3d", conv3d_plain);
    m.impl("batch_norm_relu", batch_norm_relu_plain);
    m.impl("relu_fwd", relu_fwd);
    m.impl("relu_bwd", relu_bwd);
    m.impl("linear_fwd", linear_fwd);
    m.impl("linear_dgrad", linear_dgrad);
    m.impl("linear_wgrad", linear_wgrad);
    m.impl("relu", relu_fwd);
    m.impl("linear", linear_plain);
}

TORCH_LIBRARY_IMPL(zsv, Autograd, m) {
    m.impl("conv3d", conv3d_autograd);
    m.impl("batch_norm_relu", batch_norm_relu_autograd);
    m.impl("relu", relu_autograd);
    m.impl("linear", linear_autograd);
}
