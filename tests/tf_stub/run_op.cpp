// TEST INFRASTRUCTURE: runs the WarpRNNT op of warp-transducer_amd/tensorflow_binding/warprnnt_op.cc through the TensorFlow
// stand-in of this directory.  The op source is compiled into this program unchanged (#include below); the kernel is taken
// from the registry REGISTER_KERNEL_BUILDER filled, constructed with the attr, and its Compute() is called on tensors read
// from stdin.  stdin:  device("CPU"|"GPU") blank B T U V  then B*T*U*V floats, B*(U-1) ints, B ints (input_lengths),
// B ints (label_lengths).  stdout: "status <text>", "shapes ...", B costs, B*T*U*V gradient values.
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "../../warp-transducer_amd/tensorflow_binding/warprnnt_op.cc"

namespace tf = tensorflow;

template <typename T> static tf::Tensor make(tf::DataType dt, const tf::TensorShape& s, const std::vector<T>& host, bool dev) {
    tf::Tensor t(dt, s, dev);
#ifdef TF_STUB_DEVICE_MEMORY
    if (dev) { if (!host.empty() && hipMemcpy(t.raw(), host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) std::abort(); return t; }
#endif
    if (!host.empty()) std::memcpy(t.raw(), host.data(), host.size() * sizeof(T));
    return t;
}

int main() {
    std::string device;
    int blank, B, T, U, V;
    std::cin >> device >> blank >> B >> T >> U >> V;
    std::vector<float> acts(static_cast<size_t>(B) * T * U * V);
    std::vector<int> labels(static_cast<size_t>(B) * (U - 1)), tl(B), ll(B);
    for (auto& v : acts) std::cin >> v;
    for (auto& v : labels) std::cin >> v;
    for (auto& v : tl) std::cin >> v;
    for (auto& v : ll) std::cin >> v;
    // the op as registered
    if (tf::op_registry().size() != 1 || tf::op_registry()[0].name != "WarpRNNT") { std::puts("status op not registered"); return 1; }
    const tf::OpDef& od = tf::op_registry()[0];
    std::printf("op %s inputs %zu outputs %zu attrs %zu\n", od.name.c_str(), od.inputs.size(), od.outputs.size(), od.attrs.size());
    {   // its shape function: (B,T,U,V) -> costs (B), grads (B,T,U,V)
        tf::shape_inference::InferenceContext ic;
        ic.inputs = {{{B, T, U, V}, true}, {{B, U - 1}, true}, {{B}, true}, {{B}, true}};
        const tf::Status s = od.shape_fn(&ic);
        std::printf("shapes %s costs_rank %zu grads_rank %zu\n", s.ToString().c_str(), ic.outputs[0].dims.size(), ic.outputs[1].dims.size());
        ic.inputs[0] = {{B, T, U}, true};
        std::printf("shapes_bad_rank %s\n", od.shape_fn(&ic).ok() ? "OK" : "rejected");
    }
    const tf::KernelDef* kd = nullptr;
    for (const auto& k : tf::kernel_registry())
        if (k.op == "WarpRNNT" && k.device == device) kd = &k;
    if (kd == nullptr) { std::printf("status no %s kernel registered\n", device.c_str()); return 1; }
    tf::OpKernelConstruction cons;
    cons.int_attrs["blank_label"] = blank;
    std::unique_ptr<tf::OpKernel> kernel(kd->make(&cons));
    if (!cons.status.ok()) { std::printf("status construction: %s\n", cons.status.ToString().c_str()); return 1; }
    const bool gpu = device == "GPU";
    tf::OpKernelContext ctx;
    ctx.gpu = gpu;
    ctx.input_names = {"acts", "labels", "input_lengths", "label_lengths"};
    ctx.inputs.push_back(make<float>(tf::DT_FLOAT, tf::TensorShape({B, T, U, V}), acts, gpu));
    ctx.inputs.push_back(make<int>(tf::DT_INT32, tf::TensorShape({B, U - 1}), labels, gpu));
    ctx.inputs.push_back(make<int>(tf::DT_INT32, tf::TensorShape({B}), tl, gpu));
    ctx.inputs.push_back(make<int>(tf::DT_INT32, tf::TensorShape({B}), ll, gpu));
    for (const auto& o : od.outputs) {
        bool host = false;
        for (const auto& h : kd->host_memory) host = host || o.rfind(h + ":", 0) == 0;
        ctx.output_on_host.push_back(host);
    }
    kernel->Compute(&ctx);
#ifdef TF_STUB_DEVICE_MEMORY
    if (gpu && hipDeviceSynchronize() != hipSuccess) { std::puts("status device synchronisation failed"); return 1; }
#endif
    std::printf("status %s\n", ctx.status().ToString().c_str());
    if (!ctx.status().ok()) return 0;
    for (int o = 0; o < 2; ++o) {
        const tf::Tensor& t = *ctx.outputs[o];
        std::vector<float> host(static_cast<size_t>(t.NumElements()));
#ifdef TF_STUB_DEVICE_MEMORY
        if (t.on_device()) { if (hipMemcpy(host.data(), t.raw(), host.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) std::abort(); }
        else
#endif
            std::memcpy(host.data(), t.raw(), host.size() * 4);
        for (float v : host) std::printf("%.9g\n", v);
    }
    return 0;
}
