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
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (no TensorFlow there): build with build.sh next to this file on a
// machine with tensorflow-rocm; tests/test_tensorflow_binding.py runs when `import tensorflow` works.
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

REGISTER_KERNEL_BUILDER(Name("WarpRNNT").Device(tensorflow::DEVICE_GPU), WarpRnntGpuOp);
