// TensorFlow custom ops "DAUConv" / "DAUConvGrad" on top of the C ABI (include/dau_conv.h, libdau_conv_hip.so).
//
// Same op names, input order, attribute names / types / defaults and shape function as the reference registration
// (plugins/tensorflow/src/dau_conv_op.cpp:22-84, dau_conv_grad_op.cpp:18-49), so the reference's Python package
// (dau_conv.py, _dau_conv_grad_op.py) loads this library unchanged.  NOT BUILT IN THIS REPOSITORY'S IMAGE: TensorFlow
// is not installed here; the Python surface shipped and tested in this repo is dau-convnet_amd/dau_conv (PyTorch).
//
//   TF_CFLAGS=$(python -c 'import tensorflow as tf; print(" ".join(tf.sysconfig.get_compile_flags()))')
//   TF_LFLAGS=$(python -c 'import tensorflow as tf; print(" ".join(tf.sysconfig.get_link_flags()))')
//   hipcc -std=c++17 -shared -fPIC dau_conv_ops_rocm.cc -I../../include $TF_CFLAGS $TF_LFLAGS \
//         -L../../dau-convnet_amd/dau_conv -ldau_conv_hip -o dau_conv_op.so
#include <hip/hip_runtime.h>

#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>

#include "dau_conv.h"
#include "tensorflow/core/framework/op.h"
#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/shape_inference.h"

namespace tf = tensorflow;

// attribute table shared by both ops: (name, TF type, default, C++ type)
#define DAU_ATTRS(X, BORDER_DEFAULT)                                  \
    X(number_units_x, "int", "2", int)                                \
    X(number_units_y, "int", "2", int)                                \
    X(number_units_ignore, "int", "0", int)                           \
    X(num_output, "int", "64", int)                                   \
    X(kernel_size, "int", "9", int)                                   \
    X(pad, "int", "4", int)                                           \
    X(stride, "int", "1", int)                                        \
    X(unit_normalization, "bool", "true", bool)                       \
    X(square_unit_normalization, "bool", "false", bool)               \
    X(mean_iteration_step, "int", "1", int)                           \
    X(sigma_iteration_step, "int", "1", int)                          \
    X(component_border_bound, "float", BORDER_DEFAULT, float)         \
    X(sigma_lower_bound, "float", "0.3", float)                       \
    X(merge_iteration_step, "int", "0", int)                          \
    X(merge_threshold, "int", "1", int)                               \
    X(unit_testing, "bool", "false", bool)                            \
    X(mu_learning_rate_factor, "float", "1.0", float)                 \
    X(single_dim_kernel, "bool", "false", bool)                       \
    X(forbid_positive_dim1, "bool", "false", bool)                    \
    X(use_interpolation, "bool", "true", bool)

#define DAU_ATTR_DECL(name, type, dflt, ctype) .Attr(#name ": " type " = " dflt)

// every parameter tensor is rank 4 with last dimension == num_output; output = input with dim 1 replaced by num_output
static tf::Status ParamShapes(tf::shape_inference::InferenceContext* c, int first_param, tf::shape_inference::ShapeHandle* data,
                              tf::shape_inference::DimensionHandle* outputs) {
    int num_output = 0;
    TF_RETURN_IF_ERROR(c->GetAttr("num_output", &num_output));
    TF_RETURN_IF_ERROR(c->WithRank(c->input(first_param - 1), 4, data));
    for (int i = 0; i < 4; ++i) {
        tf::shape_inference::ShapeHandle p;
        TF_RETURN_IF_ERROR(c->WithRank(c->input(first_param + i), 4, &p));
        TF_RETURN_IF_ERROR(c->WithValue(c->Dim(p, 3), num_output, outputs));
    }
    return tf::Status();
}

REGISTER_OP("DAUConv")
    .Input("input: float").Input("weights: float").Input("mu1: float").Input("mu2: float").Input("sigma: float")
    .Output("output: float")
    DAU_ATTRS(DAU_ATTR_DECL, "1")
    .SetShapeFn([](tf::shape_inference::InferenceContext* c) {
        tf::shape_inference::ShapeHandle in, out;
        tf::shape_inference::DimensionHandle f;
        TF_RETURN_IF_ERROR(ParamShapes(c, 1, &in, &f));
        TF_RETURN_IF_ERROR(c->ReplaceDim(in, 1, f, &out));
        c->set_output(0, out);
        return tf::Status();
    });

REGISTER_OP("DAUConvGrad")
    .Input("grad: float").Input("input: float").Input("weights: float").Input("mu1: float").Input("mu2: float")
    .Input("sigma: float")
    .Output("grad_input: float").Output("grad_weights: float").Output("grad_mu1: float").Output("grad_mu2: float")
    .Output("grad_sigma: float")
    DAU_ATTRS(DAU_ATTR_DECL, "0")
    .SetShapeFn([](tf::shape_inference::InferenceContext* c) {
        for (int i = 0; i < 5; ++i) c->set_output(i, c->input(i + 1));   // each gradient has the shape of its tensor
        return tf::Status();
    });

namespace {

struct Attrs {
#define DAU_ATTR_FIELD(name, type, dflt, ctype) ctype name;
    DAU_ATTRS(DAU_ATTR_FIELD, "")
#undef DAU_ATTR_FIELD
    explicit Attrs(tf::OpKernelConstruction* ctx) {
#define DAU_ATTR_READ(name, type, dflt, ctype) OP_REQUIRES_OK(ctx, ctx->GetAttr(#name, &name));
        DAU_ATTRS(DAU_ATTR_READ, "")
#undef DAU_ATTR_READ
    }
    // dau_conv_desc from the attributes and the tensor shapes of one call
    dau_conv_desc Describe(const tf::Tensor& x, const tf::Tensor& w, float sigma0) const {
        dau_conv_desc d{};
        d.struct_size = sizeof(d);
        d.batch = static_cast<int32_t>(x.dim_size(0)); d.in_channels = static_cast<int32_t>(x.dim_size(1));
        d.height = static_cast<int32_t>(x.dim_size(2)); d.width = static_cast<int32_t>(x.dim_size(3));
        d.units_per_channel = static_cast<int32_t>(w.dim_size(2)); d.out_channels = static_cast<int32_t>(w.dim_size(3));
        d.max_kernel_size = kernel_size; d.number_units_ignore = number_units_ignore;
        d.flags = (use_interpolation ? DAU_FLAG_USE_INTERPOLATION : 0) | (unit_testing ? DAU_FLAG_UNIT_TESTING : 0) |
                  (single_dim_kernel ? DAU_FLAG_SINGLE_DIM_KERNEL : 0) | (forbid_positive_dim1 ? DAU_FLAG_FORBID_POSITIVE_DIM1 : 0);
        d.algo = DAU_ALGO_AUTO; d.sigma_hint = sigma0; d.mu_learning_rate_factor = mu_learning_rate_factor;
        return d;
    }
};

tf::Status ToStatus(int rc) {
    const std::string msg = dau_conv_last_error();
    switch (rc) {
        case DAU_OK: return tf::Status();
        case DAU_INVALID_ARGUMENT: return tf::errors::InvalidArgument(msg);
        case DAU_FAILED_PRECONDITION: return tf::errors::FailedPrecondition(msg);
        default: return tf::errors::Internal(msg);
    }
}

// sigma[0], read on the host exactly as the reference's LayerSetUp does (base_dau_conv_layer.cpp:140-143)
float HostSigma(const tf::Tensor& sigma, hipStream_t st) {
    float s = 0.0f;
    (void)hipMemcpyAsync(&s, sigma.flat<float>().data(), sizeof(float), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    return s;
}

hipStream_t StreamOf(tf::OpKernelContext* ctx) {
    // the ROCm build of TensorFlow exposes the hipStream_t of the op's device context through its stream executor
    return *reinterpret_cast<hipStream_t*>(ctx->op_device_context()->stream()->platform_specific_handle().stream);
}

// Plans are kept per op instance, keyed by the descriptor: creating one is cheap, but a kept plan carries the offset-bucket
// hint from call to call (include/dau_conv.h, "Offset buckets"), which is what lets a layer with a large max_kernel_size and
// small offsets run the small-offset kernels, as the reference's per-call amax does (dau_conv_op.cpp:223-253).
// sigma enters a plan only through the prefilter support 2*ceil(5*sigma)+1, so the key holds the descriptor with sigma_hint
// replaced by the smallest sigma of that support: a trainable sigma (new value every step) keeps hitting the same plan.  The
// cache is bounded (least recently used plan dropped; an op sees a handful of shapes in its life).
// Plans are handed out reference-counted: concurrent Session::Run calls or executor threads can run Compute of one kernel at the
// same time, so a Compute may still be inside dau_conv_forward / dau_conv_check_status with a plan while another Compute's new
// shape evicts it from the cache.  Eviction only drops the cache's reference; the plan (and its pinned status block) is destroyed
// when the last Compute that holds it returns.
using PlanRef = std::shared_ptr<dau_conv_plan>;
class PlanCache {
  public:
    static constexpr size_t kMaxPlans = 16;
    int Get(const dau_conv_desc& d, PlanRef* plan) {
        dau_conv_desc canon = d;
        canon.sigma_hint = static_cast<float>(dau_conv_filter_support(d.sigma_hint));   // the support stands in for sigma
        const std::string key(reinterpret_cast<const char*>(&canon), sizeof(canon));
        std::lock_guard<std::mutex> lock(mu_);
        for (auto it = plans_.begin(); it != plans_.end(); ++it)
            if (it->first == key) {
                plans_.splice(plans_.begin(), plans_, it);                               // most recently used first
                *plan = plans_.front().second;
                return DAU_OK;
            }
        dau_conv_plan* raw = nullptr;
        const int rc = dau_conv_plan_create(&d, &raw);
        if (rc != DAU_OK) return rc;
        plan->reset(raw, [](dau_conv_plan* p) { dau_conv_plan_destroy(p); });
        plans_.emplace_front(key, *plan);
        while (plans_.size() > kMaxPlans) plans_.pop_back();                             // drops the cache's reference only
        return DAU_OK;
    }
  private:
    std::mutex mu_;
    std::list<std::pair<std::string, PlanRef>> plans_;
};

class DAUConvOp : public tf::OpKernel {
  public:
    explicit DAUConvOp(tf::OpKernelConstruction* ctx) : tf::OpKernel(ctx), attrs_(ctx) {}
    void Compute(tf::OpKernelContext* ctx) override {
        const tf::Tensor &x = ctx->input(0), &w = ctx->input(1), &mu1 = ctx->input(2), &mu2 = ctx->input(3), &sigma = ctx->input(4);
        OP_REQUIRES(ctx, x.dims() == 4 && w.dims() == 4, tf::errors::InvalidArgument("input and parameters must have rank 4"));
        hipStream_t st = StreamOf(ctx);
        const dau_conv_desc d = attrs_.Describe(x, w, HostSigma(sigma, st));
        PlanRef held;                                            // keeps the plan alive for the whole call (see PlanCache)
        OP_REQUIRES_OK(ctx, ToStatus(plans_.Get(d, &held)));
        dau_conv_plan* plan = held.get();
        tf::Tensor* y = nullptr;
        OP_REQUIRES_OK(ctx, ctx->allocate_output(0, tf::TensorShape({d.batch, d.out_channels, d.height, d.width}), &y));
        size_t ws_bytes = 0;
        OP_REQUIRES_OK(ctx, ToStatus(dau_conv_workspace_bytes(plan, DAU_PASS_FORWARD, &ws_bytes)));
        tf::Tensor ws;
        OP_REQUIRES_OK(ctx, ctx->allocate_temp(tf::DT_INT8, tf::TensorShape({static_cast<tf::int64>(ws_bytes)}), &ws));
        int rc = dau_conv_forward(plan, st, x.flat<float>().data(), w.flat<float>().data(), mu1.flat<float>().data(),
                                  mu2.flat<float>().data(), sigma.flat<float>().data(), y->flat<float>().data(),
                                  ws.flat<tf::int8>().data(), ws_bytes);
        if (rc == DAU_OK) rc = dau_conv_check_status(plan, st, ws.flat<tf::int8>().data(), nullptr);   // NaN / out-of-range offsets
        OP_REQUIRES_OK(ctx, ToStatus(rc));
    }
  private:
    Attrs attrs_;
    PlanCache plans_;
};

class DAUConvGradOp : public tf::OpKernel {
  public:
    explicit DAUConvGradOp(tf::OpKernelConstruction* ctx) : tf::OpKernel(ctx), attrs_(ctx) {}
    void Compute(tf::OpKernelContext* ctx) override {
        const tf::Tensor &dy = ctx->input(0), &x = ctx->input(1), &w = ctx->input(2), &mu1 = ctx->input(3), &mu2 = ctx->input(4),
                         &sigma = ctx->input(5);
        hipStream_t st = StreamOf(ctx);
        const dau_conv_desc d = attrs_.Describe(x, w, HostSigma(sigma, st));
        PlanRef held;                                            // keeps the plan alive for the whole call (see PlanCache)
        OP_REQUIRES_OK(ctx, ToStatus(plans_.Get(d, &held)));
        dau_conv_plan* plan = held.get();
        tf::Tensor* out[5];
        for (int i = 0; i < 5; ++i) OP_REQUIRES_OK(ctx, ctx->allocate_output(i, ctx->input(i + 1).shape(), &out[i]));
        size_t ws_bytes = 0;
        OP_REQUIRES_OK(ctx, ToStatus(dau_conv_workspace_bytes(plan, DAU_PASS_BACKWARD, &ws_bytes)));
        tf::Tensor ws;
        OP_REQUIRES_OK(ctx, ctx->allocate_temp(tf::DT_INT8, tf::TensorShape({static_cast<tf::int64>(ws_bytes)}), &ws));
        // the ABI overwrites the gradients (the reference zero-fills and accumulates, dau_conv_grad_op.cpp:202-205)
        int rc = dau_conv_backward(plan, st, x.flat<float>().data(), dy.flat<float>().data(), w.flat<float>().data(),
                                   mu1.flat<float>().data(), mu2.flat<float>().data(), sigma.flat<float>().data(),
                                   out[0]->flat<float>().data(), out[1]->flat<float>().data(), out[2]->flat<float>().data(),
                                   out[3]->flat<float>().data(), out[4]->flat<float>().data(), ws.flat<tf::int8>().data(),
                                   ws_bytes, DAU_NEED_ALL);
        if (rc == DAU_OK) rc = dau_conv_check_status(plan, st, ws.flat<tf::int8>().data(), nullptr);
        OP_REQUIRES_OK(ctx, ToStatus(rc));
    }
  private:
    Attrs attrs_;
    PlanCache plans_;
};

}  // namespace

REGISTER_KERNEL_BUILDER(Name("DAUConv").Device(tf::DEVICE_GPU), DAUConvOp);
REGISTER_KERNEL_BUILDER(Name("DAUConvGrad").Device(tf::DEVICE_GPU), DAUConvGradOp);
