// TEST INFRASTRUCTURE (see op_kernel.h in this directory): REGISTER_OP and the shape-inference surface.
#pragma once
#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/shape_inference.h"

namespace tensorflow {
struct OpDef {
    std::string name;
    std::vector<std::string> inputs, outputs, attrs;
    std::function<Status(shape_inference::InferenceContext*)> shape_fn;
};
inline std::vector<OpDef>& op_registry() { static std::vector<OpDef> r; return r; }
class OpDefBuilder {
public:
    explicit OpDefBuilder(const char* name) { def_.name = name; }
    OpDefBuilder& Input(const char* s) { def_.inputs.push_back(s); return *this; }
    OpDefBuilder& Output(const char* s) { def_.outputs.push_back(s); return *this; }
    OpDefBuilder& Attr(const char* s) { def_.attrs.push_back(s); return *this; }
    OpDefBuilder& Doc(const char*) { return *this; }
    OpDefBuilder& SetShapeFn(std::function<Status(shape_inference::InferenceContext*)> f) { def_.shape_fn = std::move(f); return *this; }
    OpDef def_;
};
struct OpRegistrar { OpRegistrar(const OpDefBuilder& b) { op_registry().push_back(b.def_); } };
#define REGISTER_OP(NAME) static ::tensorflow::OpRegistrar TF_STUB_CAT(tf_stub_op_, __COUNTER__) = ::tensorflow::OpDefBuilder(NAME)
}  // namespace tensorflow
