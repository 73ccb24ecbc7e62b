// TEST INFRASTRUCTURE (see op_kernel.h in this directory): the InferenceContext calls a shape function makes.
#pragma once
#include "tensorflow/core/framework/op_kernel.h"

namespace tensorflow {
namespace shape_inference {
struct DimensionHandle { int64 v = -1; };
struct ShapeHandle { std::vector<int64> dims; bool known = false; };
class InferenceContext {
public:
    std::vector<ShapeHandle> inputs, outputs;
    ShapeHandle input(int i) const { return inputs[i]; }
    Status WithRank(ShapeHandle s, int64 rank, ShapeHandle* out) {
        if (s.known && static_cast<int64>(s.dims.size()) != rank) return errors::InvalidArgument("Shape must be rank ", rank, " but is rank ", s.dims.size());
        *out = s;
        return Status();
    }
    DimensionHandle Dim(ShapeHandle s, int i) const { return {s.known ? s.dims[i] : -1}; }
    ShapeHandle Vector(DimensionHandle d) const { return {{d.v}, true}; }
    void set_output(int i, ShapeHandle s) { if (static_cast<int>(outputs.size()) <= i) outputs.resize(i + 1); outputs[i] = s; }
};
}  // namespace shape_inference
}  // namespace tensorflow
