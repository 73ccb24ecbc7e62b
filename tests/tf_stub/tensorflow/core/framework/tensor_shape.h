// TEST INFRASTRUCTURE (see op_kernel.h in this directory)
#pragma once
#include "tensorflow/core/framework/op_kernel.h"
