// tanh-approximated gelu and its derivative, evaluated in the order of the reference's expression (examples/bert.py:12):
//   0.5 * x * (1.0 + (x * 0.7978845608 * (1.0 + 0.044715 * x * x)).tanh())
// One definition for the elementwise kernels and the GEMM epilogues: the same bits either way.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

namespace lg {

__device__ __forceinline__ float gelu_inner(float x) { return (x * 0.7978845608f) * (1.0f + (0.044715f * x) * x); }
__device__ __forceinline__ float gelu_value(float x) { return (0.5f * x) * (1.0f + tanhf(gelu_inner(x))); }
// g * d/dx [0.5 x (1 + tanh u)] = g * (0.5 (1 + tanh u) + 0.5 x (1 - tanh^2 u) u'),  u' = 0.7978845608 (1 + 3*0.044715 x^2)
__device__ __forceinline__ float gelu_grad(float x, float g) {
    const float th = tanhf(gelu_inner(x));
    const float du = 0.7978845608f * (1.0f + 0.134145f * x * x);
    return g * (0.5f * (1.0f + th) + (0.5f * x) * (1.0f - th * th) * du);
}

}  // namespace lg
