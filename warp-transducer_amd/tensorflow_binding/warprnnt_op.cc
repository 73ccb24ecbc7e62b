// TensorFlow (ROCm build) custom op "WarpRNNT" over the C-ABI of include/rnnt.h.
//
// Same op contract as the reference's second binding (tensorflow_binding/src/warprnnt_op.cc:13-20):
//   WarpRNNT(acts f32 (B,T,U,V), labels i32 (B,U-1), input_lengths i32 (B), label_lengths i32 (B);
//            blank_label: int = 0) -> (costs f32 (B), grads f32 (B,T,U,V))
// so `warprnnt_tensorflow.rnnt_loss` and its registered gradient keep working unchanged.  Differences, all
// on the inside: the kernel calls compute_rnnt_loss_async -- everything is enqueued on the op's HIP stream,
// costs are written on the device (the reference pins `costs` to host memory and synchronises the stream
// inside Compute, warprnnt_op.cc:185-187), and the workspace is a TF temporary.
//
// A CPU kernel is registered too, as in the reference (tensorflow_binding/src/warprnnt_op.cc:142-161): RNNT_CPU location,
// log-probabilities in (the Python wrapper applies tf.nn.log_softmax on the CPU, tensorflow_binding/tests/test_warprnnt_op.py:20),
// sparse log-prob gradients out, threads from the device's worker pool.
//
// This repository's image has no TensorFlow.  What HAS happened to this file here: it is compiled against a stand-in for the
// TensorFlow declarations it uses (tests/tf_stub/, test infrastructure) and both kernels' Compute() are executed through that
// stand-in on the reference's golden vectors -- the CPU kernel everywhere, the GPU kernel on the MI355X
// (tests/test_tensorflow_stub.py).  Against the REAL headers: build with build.sh next to this file on a machine with
// tensorflow-rocm; tests/test_tensorflow_binding.py runs when `import tensorflow` works.
#define EIGEN_USE_GPU 1

#include "tensorflow/core/framework/op.h"
#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/shape_inference.h"
#include "tensorflow/core/framework/tensor.h"
#include "tensorflow/core/framework/tensor_shape.h"

#include "rnnt.h"

namespace {

namespace tf = tensorflow;

tf::Status RnntShapes(tf::shape_inference::InferenceContext* c) {
    tf::shape_inference::ShapeHandle acts, unused;
    TF_RETURN_IF_ERROR(c->WithRank(c->input(0), 4, &acts));
    TF_RETURN_IF_ERROR(c->WithRank(c->input(1), 2, &unused));
    TF_RETURN_IF_ERROR(c->WithRank(c->input(2), 1, &unused));
    TF_RETURN_IF_ERROR(c->WithRank(c->input(3), 1, &unused));
    c->set_output(0, c->Vector(c->Dim(acts, 0)));
    c->set_output(1, acts);
    return tf::Status();
}

class WarpRnntGpuOp : public tf::OpKernel {
public:
    explicit WarpRnntGpuOp(tf::OpKernelConstruction* ctx) : tf::OpKernel(ctx) {
        OP_REQUIRES_OK(ctx, ctx->GetAttr("blank_label", &blank_));
    }

    void Compute(tf::OpKernelContext* ctx) override {
        const tf::Tensor& acts = ctx->input(0);
        const tf::Tensor& labels = ctx->input(1);
        const tf::Tensor& input_lengths = ctx->input(2);
        const tf::Tensor& label_lengths = ctx->input(3);
        OP_REQUIRES(ctx, acts.dims() == 4, tf::errors::InvalidArgument("acts must be (B, T, U, V)"));
        const tf::int64 B = acts.dim_size(0), T = acts.dim_size(1), U = acts.dim_size(2), V = acts.dim_size(3);
        OP_REQUIRES(ctx, labels.dims() == 2 && labels.dim_size(0) == B && labels.dim_size(1) == U - 1,
                    tf::errors::InvalidArgument("labels must be (B, U-1)"));
        OP_REQUIRES(ctx, input_lengths.dims() == 1 && input_lengths.dim_size(0) == B &&
                             label_lengths.dims() == 1 && label_lengths.dim_size(0) == B,
                    tf::errors::InvalidArgument("must have a length per example"));

        tf::Tensor* costs = nullptr;
        tf::Tensor* grads = nullptr;
        OP_REQUIRES_OK(ctx, ctx->allocate_output(0, tf::TensorShape({B}), &costs));
        OP_REQUIRES_OK(ctx, ctx->allocate_output(1, acts.shape(), &grads));

        size_t bytes = 0;
        rnntStatus_t st = get_workspace_size(static_cast<int>(T), static_cast<int>(U), static_cast<int>(B), true,
                                             &bytes, sizeof(float));
        OP_REQUIRES(ctx, st == RNNT_STATUS_SUCCESS,
                    tf::errors::Internal("get_workspace_size: ", rnntGetStatusString(st)));
        tf::Tensor workspace;
        OP_REQUIRES_OK(ctx, ctx->allocate_temp(tf::DT_UINT8, tf::TensorShape({static_cast<tf::int64>(bytes)}),
                                               &workspace));

        rnntOptions options{};
        options.loc = RNNT_GPU;
        options.blank_label = blank_;
        options.maxT = static_cast<int>(T);
        options.maxU = static_cast<int>(U);
        options.batch_first = true;
        options.stream = reinterpret_cast<CUstream>(ctx->eigen_device<Eigen::GpuDevice>().stream());

        // U == 1: there are no labels and the tensor is empty; the library wants a non-null pointer it never reads
        const void* label_ptr = labels.NumElements() ? static_cast<const void*>(labels.flat<tf::int32>().data())
                                                     : static_cast<const void*>(costs->flat<float>().data());
        st = compute_rnnt_loss_async(acts.flat<float>().data(), grads->flat<float>().data(),
                                     static_cast<const int*>(label_ptr), label_lengths.flat<tf::int32>().data(),
                                     input_lengths.flat<tf::int32>().data(), static_cast<int>(V), static_cast<int>(B),
                                     costs->flat<float>().data(), /*grad_scale_device=*/nullptr,
                                     workspace.flat<tf::uint8>().data(), options, /*dtype_code=*/0);
        OP_REQUIRES(ctx, st == RNNT_STATUS_SUCCESS,
                    tf::errors::Internal("compute_rnnt_loss_async: ", rnntGetStatusString(st)));
    }

private:
    int blank_ = 0;
};

// The reference's CPU kernel (tensorflow_binding/src/warprnnt_op.cc:30-161 with create_options of :150-156): every tensor on the
// host, activations are LOG-PROBABILITIES, the second output is the sparse d(cost)/d(log-probs) of include/rnnt.h's RNNT_CPU
// contract; the call returns when the result is there (nothing asynchronous on this device).
class WarpRnntCpuOp : public tf::OpKernel {
public:
    explicit WarpRnntCpuOp(tf::OpKernelConstruction* ctx) : tf::OpKernel(ctx) {
        OP_REQUIRES_OK(ctx, ctx->GetAttr("blank_label", &blank_));
    }

    void Compute(tf::OpKernelContext* ctx) override {
        const tf::Tensor& acts = ctx->input(0);
        const tf::Tensor& labels = ctx->input(1);
        const tf::Tensor& input_lengths = ctx->input(2);
        const tf::Tensor& label_lengths = ctx->input(3);
        OP_REQUIRES(ctx, acts.dims() == 4, tf::errors::InvalidArgument("acts is not a 4-Tensor"));
        OP_REQUIRES(ctx, labels.dims() == 2, tf::errors::InvalidArgument("labels is not a 2-Tensor"));
        const tf::int64 B = acts.dim_size(0), T = acts.dim_size(1), U = acts.dim_size(2), V = acts.dim_size(3);
        OP_REQUIRES(ctx, labels.dim_size(0) == B && labels.dim_size(1) == U - 1, tf::errors::InvalidArgument("labels must be (B, U-1)"));
        OP_REQUIRES(ctx, input_lengths.dims() == 1 && input_lengths.dim_size(0) == B,
                    tf::errors::InvalidArgument("input_lengths is not a vector of one length per example"));
        OP_REQUIRES(ctx, label_lengths.dims() == 1 && label_lengths.dim_size(0) == B,
                    tf::errors::InvalidArgument("label_lengths is not a vector of one length per example"));
        tf::Tensor* costs = nullptr;
        tf::Tensor* grads = nullptr;
        OP_REQUIRES_OK(ctx, ctx->allocate_output(0, tf::TensorShape({B}), &costs));
        OP_REQUIRES_OK(ctx, ctx->allocate_output(1, acts.shape(), &grads));
        size_t bytes = 0;
        rnntStatus_t st = get_workspace_size(static_cast<int>(T), static_cast<int>(U), static_cast<int>(B), false, &bytes, sizeof(float));
        OP_REQUIRES(ctx, st == RNNT_STATUS_SUCCESS, tf::errors::Internal("warp_rnnt error in get_workspace_size: ", rnntGetStatusString(st)));
        tf::Tensor workspace;
        OP_REQUIRES_OK(ctx, ctx->allocate_temp(tf::DT_UINT8, tf::TensorShape({static_cast<tf::int64>(bytes)}), &workspace));
        rnntOptions options{};
        options.loc = RNNT_CPU;
        options.batch_first = true;
        options.blank_label = blank_;
        options.maxT = static_cast<int>(T);
        options.maxU = static_cast<int>(U);
        options.num_threads = static_cast<unsigned>(ctx->device()->tensorflow_cpu_worker_threads()->num_threads);
        const void* label_ptr = labels.NumElements() ? static_cast<const void*>(labels.flat<tf::int32>().data())
                                                     : static_cast<const void*>(costs->flat<float>().data());
        st = compute_rnnt_loss(acts.flat<float>().data(), grads->flat<float>().data(), static_cast<const int*>(label_ptr),
                               label_lengths.flat<tf::int32>().data(), input_lengths.flat<tf::int32>().data(), static_cast<int>(V),
                               static_cast<int>(B), costs->flat<float>().data(), workspace.flat<tf::uint8>().data(), options);
        OP_REQUIRES(ctx, st == RNNT_STATUS_SUCCESS, tf::errors::Internal("warp_rnnt error in compute_rnnt_loss: ", rnntGetStatusString(st)));
    }

private:
    int blank_ = 0;
};

}  // namespace

REGISTER_OP("WarpRNNT")
    .Input("acts: float32")
    .Input("labels: int32")
    .Input("input_lengths: int32")
    .Input("label_lengths: int32")
    .Attr("blank_label: int = 0")
    .Output("costs: float32")
    .Output("grads: float32")
    .SetShapeFn(RnntShapes);

REGISTER_KERNEL_BUILDER(Name("WarpRNNT").Device(tensorflow::DEVICE_CPU), WarpRnntCpuOp);
REGISTER_KERNEL_BUILDER(Name("WarpRNNT").Device(tensorflow::DEVICE_GPU), WarpRnntGpuOp);
