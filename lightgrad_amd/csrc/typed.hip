// Arithmetic on tensors that are NOT float32 (include/lghip.h: lg_ew_typed / lg_reduce_typed / lg_cast): int16, int32, int64
// and float64 - the dtypes the reference's tensors can hold besides float32 (cpu/tensor.py:45-46 keeps the array's dtype:
// MNIST labels are int16, data.py:43; BERT ids int32, examples/bert.py:346) and that its device backend generates every
// elementwise / reduction kernel for (opencl/kernels.py:9, :24-107 `dtype_to_ctype`, :344-431).  Numeric ground truth is numpy
// (cpu/ops.py:52-84, :260-293): integer arithmetic wraps around (two's complement), sums of integers are formed in int64 (numpy's
// default accumulator for narrower signed integers), max / min keep the dtype and propagate NaN, float64 is plain IEEE double.
//
// The north-star path is float32 (elementwise.hip / reduce.hip: float4 paths, tuned grids); these kernels make the other dtypes
// CORRECT on the device, not fast: one element per thread through the collapsed stride descriptor for elementwise work and casts,
// one workgroup per output element for reductions.  Memory-bound all the same: 1 - 8 byte items, coalesced for dense operands.
#include "common.h"
#include <type_traits>

namespace lg {

template <typename T> struct Wrap { using U = std::make_unsigned_t<T>; };
template <> struct Wrap<double> { using U = double; };

// a (op) b in T with numpy's semantics; integers through their unsigned twins (no signed-overflow UB, the same bits)
template <typename T, int OP>
__device__ __forceinline__ T typed_apply(T a, T b) {
    using U = typename Wrap<T>::U;
    if constexpr (OP == LG_EW_COPY) return a;
    else if constexpr (OP == LG_EW_NEG) return std::is_same_v<T, double> ? T(-a) : T(U(0) - U(a));
    else if constexpr (OP == LG_EW_ADD) return T(U(a) + U(b));
    else if constexpr (OP == LG_EW_SUB) return T(U(a) - U(b));
    else if constexpr (OP == LG_EW_MUL) return T(U(a) * U(b));
    else if constexpr (OP == LG_EW_DIV) return a / b;            // float64 only (the host refuses integers: numpy gives float64 there)
    else return T(pow(double(a), double(b)));                    // LG_EW_POW, float64 only
}

struct TypedArgs {
    void*       out;
    const void* a;        // NULL: the scalar
    const void* b;        // NULL: the scalar (binary ops) / unused (unary ops)
    double      scalar_f;
    int64_t     scalar_i;
};

// slots of IterDesc::stride: 0 = out, 2 = a, 3 = b (as lg_ew)
template <typename T, int OP, bool BINARY>
__global__ void __launch_bounds__(256) typed_ew(TypedArgs p, IterDesc d) {
    const int nd = d.ndim;
    const T s = std::is_same_v<T, double> ? T(p.scalar_f) : T(p.scalar_i);
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < d.numel; e += stride) {
        int64_t rem = e, oo = 0, oa = 0, ob = 0;
        for (int k = nd - 1; k >= 0; --k) {
            const int64_t i = rem % d.shape[k];
            rem /= d.shape[k];
            oo += i * d.stride[0][k];
            oa += i * d.stride[2][k];
            ob += i * d.stride[3][k];
        }
        const T x = p.a ? static_cast<const T*>(p.a)[oa] : s;
        const T y = BINARY ? (p.b ? static_cast<const T*>(p.b)[ob] : s) : T(0);
        static_cast<T*>(p.out)[oo] = typed_apply<T, OP>(x, y);
    }
}

template <typename S, typename D>
__global__ void __launch_bounds__(256) typed_cast(const S* __restrict__ in, D* __restrict__ out, IterDesc d) {
    const int nd = d.ndim;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < d.numel; e += stride) {
        int64_t rem = e, oo = 0, oi = 0;
        for (int k = nd - 1; k >= 0; --k) {
            const int64_t i = rem % d.shape[k];
            rem /= d.shape[k];
            oo += i * d.stride[0][k];
            oi += i * d.stride[2][k];
        }
        out[oo] = D(in[oi]);            // C++ conversion = numpy's astype for these types (float -> int truncates toward zero)
    }
}

// ---- reductions: one workgroup per output element --------------------------------------------------------------------------
struct TypedRed {
    int     nk, nr;
    int64_t kshape[LG_MAX_DIMS], kstride[LG_MAX_DIMS];
    int64_t rshape[LG_MAX_DIMS], rstride[LG_MAX_DIMS];
    int64_t rlen;
};

template <typename A, int OP>
__device__ __forceinline__ A red_combine(A a, A b) {
    if constexpr (OP == LG_RED_SUM) {
        if constexpr (std::is_same_v<A, double>) return a + b;
        else return A((unsigned long long)a + (unsigned long long)b);
    } else if constexpr (std::is_same_v<A, double>) {
        if (a != a) return a;                      // np.max / np.min propagate NaN
        if (b != b) return b;
        return OP == LG_RED_MAX ? (a > b ? a : b) : (a < b ? a : b);
    } else {
        return OP == LG_RED_MAX ? (a > b ? a : b) : (a < b ? a : b);
    }
}

// T: element type, A: accumulator (and output) type: int64 for integer sums, T otherwise
template <typename T, typename A, int OP>
__global__ void __launch_bounds__(256) typed_reduce(const T* __restrict__ in, A* __restrict__ out, TypedRed d) {
    __shared__ A red[256];
    int64_t rem = blockIdx.x, base = 0;
    for (int k = d.nk - 1; k >= 0; --k) {
        base += (rem % d.kshape[k]) * d.kstride[k];
        rem /= d.kshape[k];
    }
    A acc = A(0);
    bool any = false;
    for (int64_t i = threadIdx.x; i < d.rlen; i += 256) {
        int64_t r = i, off = base;
        for (int k = d.nr - 1; k >= 0; --k) {
            off += (r % d.rshape[k]) * d.rstride[k];
            r /= d.rshape[k];
        }
        const A v = A(in[off]);
        acc = any ? red_combine<A, OP>(acc, v) : v;
        any = true;
    }
    // threads beyond the reduction's length hold nothing: the tree below only combines slots that hold a value
    red[threadIdx.x] = acc;
    __syncthreads();
    const int64_t live = d.rlen < 256 ? d.rlen : 256;
    for (int w = 128; w > 0; w >>= 1) {
        if (int(threadIdx.x) < w && int64_t(threadIdx.x) + w < live) red[threadIdx.x] = red_combine<A, OP>(red[threadIdx.x], red[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

template <typename T>
static int launch_typed_ew(int op, bool binary, const TypedArgs& p, const IterDesc& d) {
    const dim3 grid(stream_grid(d.numel)), block(256);
    hipStream_t s = rt().stream;
    constexpr bool F = std::is_same_v<T, double>;
    switch (op) {
        case LG_EW_COPY: hipLaunchKernelGGL((typed_ew<T, LG_EW_COPY, false>), grid, block, 0, s, p, d); return LG_OK;
        case LG_EW_NEG:  hipLaunchKernelGGL((typed_ew<T, LG_EW_NEG, false>), grid, block, 0, s, p, d); return LG_OK;
        case LG_EW_ADD:  hipLaunchKernelGGL((typed_ew<T, LG_EW_ADD, true>), grid, block, 0, s, p, d); return LG_OK;
        case LG_EW_SUB:  hipLaunchKernelGGL((typed_ew<T, LG_EW_SUB, true>), grid, block, 0, s, p, d); return LG_OK;
        case LG_EW_MUL:  hipLaunchKernelGGL((typed_ew<T, LG_EW_MUL, true>), grid, block, 0, s, p, d); return LG_OK;
        case LG_EW_DIV:
            if constexpr (F) { hipLaunchKernelGGL((typed_ew<T, LG_EW_DIV, true>), grid, block, 0, s, p, d); return LG_OK; }
            break;
        case LG_EW_POW:
            if constexpr (F) { hipLaunchKernelGGL((typed_ew<T, LG_EW_POW, true>), grid, block, 0, s, p, d); return LG_OK; }
            break;
        default: break;
    }
    (void)binary;
    set_error("lg_ew_typed: op %d is not defined for this dtype (integers: copy, neg, add, sub, mul; float64 also div, pow)", op);
    return LG_EINVAL;
}

template <typename T, typename A>
static void launch_typed_reduce(int op, const void* in, void* out, const TypedRed& d, int64_t n_out) {
    const dim3 grid{unsigned(n_out)}, block(256);
    hipStream_t s = rt().stream;
    const T* src = static_cast<const T*>(in);
    if (op == LG_RED_SUM)      hipLaunchKernelGGL((typed_reduce<T, A, LG_RED_SUM>), grid, block, 0, s, src, static_cast<A*>(out), d);
    else if (op == LG_RED_MAX) hipLaunchKernelGGL((typed_reduce<T, T, LG_RED_MAX>), grid, block, 0, s, src, static_cast<T*>(out), d);
    else                       hipLaunchKernelGGL((typed_reduce<T, T, LG_RED_MIN>), grid, block, 0, s, src, static_cast<T*>(out), d);
}

template <typename S>
static int launch_cast_from(int dst, const void* in, void* out, const IterDesc& d) {
    const dim3 grid(stream_grid(d.numel)), block(256);
    hipStream_t s = rt().stream;
    const S* src = static_cast<const S*>(in);
    switch (dst) {
        case LG_DT_I16: hipLaunchKernelGGL((typed_cast<S, int16_t>), grid, block, 0, s, src, static_cast<int16_t*>(out), d); return LG_OK;
        case LG_DT_I32: hipLaunchKernelGGL((typed_cast<S, int32_t>), grid, block, 0, s, src, static_cast<int32_t*>(out), d); return LG_OK;
        case LG_DT_I64: hipLaunchKernelGGL((typed_cast<S, int64_t>), grid, block, 0, s, src, static_cast<int64_t*>(out), d); return LG_OK;
        case LG_DT_F64: hipLaunchKernelGGL((typed_cast<S, double>), grid, block, 0, s, src, static_cast<double*>(out), d); return LG_OK;
        case LG_DT_F32: hipLaunchKernelGGL((typed_cast<S, float>), grid, block, 0, s, src, static_cast<float*>(out), d); return LG_OK;
        default: break;
    }
    set_error("lg_cast: unknown destination dtype %d", dst);
    return LG_EINVAL;
}

}  // namespace lg

using namespace lg;

extern "C" int lg_ew_typed(int op, int dtype, int ndim, const int64_t* shape, void* out, const int64_t* out_strides,
                           const void* a, const int64_t* a_strides, const void* b, const int64_t* b_strides,
                           double scalar_f, int64_t scalar_i) {
    LG_REQUIRE_INIT();
    LG_ARG(ndim >= 0 && ndim <= LG_MAX_DIMS, "lg_ew_typed: ndim %d out of range [0, %d]", ndim, LG_MAX_DIMS);
    LG_ARG(out != nullptr && (ndim == 0 || (shape && out_strides)), "lg_ew_typed: NULL output / shape");
    const bool binary = !(op == LG_EW_COPY || op == LG_EW_NEG);
    LG_ARG(a != nullptr || (binary && b != nullptr), "lg_ew_typed: at least one operand must be a tensor");
    LG_ARG((a == nullptr || ndim == 0 || a_strides) && (b == nullptr || ndim == 0 || b_strides), "lg_ew_typed: operand without strides");
    const int64_t* strides[kMaxOps] = {out_strides, nullptr, a ? a_strides : nullptr, (binary && b) ? b_strides : nullptr, nullptr, nullptr};
    IterDesc d;
    LG_ARG(build_iter(ndim, shape, strides, kMaxOps, d), "lg_ew_typed: bad shape");
    for (int k = 0; k < d.ndim; ++k) LG_ARG(d.shape[k] == 1 || d.stride[0][k] != 0, "lg_ew_typed: the output has a zero stride over an extent > 1");
    if (d.numel == 0) return LG_OK;
    TypedArgs p{out, a, binary ? b : nullptr, scalar_f, scalar_i};
    int rc;
    switch (dtype) {
        case LG_DT_I16: rc = launch_typed_ew<int16_t>(op, binary, p, d); break;
        case LG_DT_I32: rc = launch_typed_ew<int32_t>(op, binary, p, d); break;
        case LG_DT_I64: rc = launch_typed_ew<int64_t>(op, binary, p, d); break;
        case LG_DT_F64: rc = launch_typed_ew<double>(op, binary, p, d); break;
        default: set_error("lg_ew_typed: dtype %d (int16 / int32 / int64 / float64; float32 goes through lg_ew)", dtype); return LG_EINVAL;
    }
    if (rc != LG_OK) return rc;
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_reduce_typed(int op, int dtype, int ndim, const int64_t* shape, const void* in, const int64_t* in_strides,
                               uint32_t axis_mask, void* out) {
    LG_REQUIRE_INIT();
    LG_ARG(ndim >= 0 && ndim <= LG_MAX_DIMS, "lg_reduce_typed: ndim %d out of range [0, %d]", ndim, LG_MAX_DIMS);
    LG_ARG(in != nullptr && out != nullptr && (ndim == 0 || (shape && in_strides)), "lg_reduce_typed: NULL pointer");
    LG_ARG(op == LG_RED_SUM || op == LG_RED_MAX || op == LG_RED_MIN, "lg_reduce_typed: unknown op id %d", op);
    LG_ARG((axis_mask >> ndim) == 0, "lg_reduce_typed: axis_mask 0x%x names a dimension >= ndim %d", axis_mask, ndim);
    TypedRed d{};
    int64_t n_out = 1;
    d.rlen = 1;
    for (int k = 0; k < ndim; ++k) {
        LG_ARG(shape[k] >= 0, "lg_reduce_typed: negative extent");
        if ((axis_mask >> k) & 1u) { d.rshape[d.nr] = shape[k]; d.rstride[d.nr] = in_strides[k]; ++d.nr; d.rlen *= shape[k]; }
        else                       { d.kshape[d.nk] = shape[k]; d.kstride[d.nk] = in_strides[k]; ++d.nk; n_out *= shape[k]; }
    }
    if (n_out == 0) return LG_OK;
    LG_ARG(n_out < (int64_t(1) << 31), "lg_reduce_typed: too many outputs for one launch");
    if (d.rlen == 0) {
        LG_ARG(op == LG_RED_SUM, "lg_reduce_typed: zero-size reduction has no identity for max/min");
        int64_t one = 1;
        return lg_fill_strided(8, 1, &n_out, out, &one, 0);          // integer sums are int64, float64 sums 8 bytes too
    }
    switch (dtype) {
        case LG_DT_I16: launch_typed_reduce<int16_t, int64_t>(op, in, out, d, n_out); break;
        case LG_DT_I32: launch_typed_reduce<int32_t, int64_t>(op, in, out, d, n_out); break;
        case LG_DT_I64: launch_typed_reduce<int64_t, int64_t>(op, in, out, d, n_out); break;
        case LG_DT_F64: launch_typed_reduce<double, double>(op, in, out, d, n_out); break;
        default: set_error("lg_reduce_typed: dtype %d (int16 / int32 / int64 / float64; float32 goes through lg_reduce)", dtype); return LG_EINVAL;
    }
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_cast(int src_dtype, int dst_dtype, int ndim, const int64_t* shape, void* out, const int64_t* out_strides,
                       const void* in, const int64_t* in_strides) {
    LG_REQUIRE_INIT();
    LG_ARG(ndim >= 0 && ndim <= LG_MAX_DIMS, "lg_cast: ndim %d out of range [0, %d]", ndim, LG_MAX_DIMS);
    LG_ARG(in != nullptr && out != nullptr && (ndim == 0 || (shape && out_strides && in_strides)), "lg_cast: NULL pointer");
    const int64_t* strides[kMaxOps] = {out_strides, nullptr, in_strides, nullptr, nullptr, nullptr};
    IterDesc d;
    LG_ARG(build_iter(ndim, shape, strides, kMaxOps, d), "lg_cast: bad shape");
    for (int k = 0; k < d.ndim; ++k) LG_ARG(d.shape[k] == 1 || d.stride[0][k] != 0, "lg_cast: the output has a zero stride over an extent > 1");
    if (d.numel == 0) return LG_OK;
    int rc;
    switch (src_dtype) {
        case LG_DT_I16: rc = launch_cast_from<int16_t>(dst_dtype, in, out, d); break;
        case LG_DT_I32: rc = launch_cast_from<int32_t>(dst_dtype, in, out, d); break;
        case LG_DT_I64: rc = launch_cast_from<int64_t>(dst_dtype, in, out, d); break;
        case LG_DT_F64: rc = launch_cast_from<double>(dst_dtype, in, out, d); break;
        case LG_DT_F32: rc = launch_cast_from<float>(dst_dtype, in, out, d); break;
        default: set_error("lg_cast: unknown source dtype %d", src_dtype); return LG_EINVAL;
    }
    if (rc != LG_OK) return rc;
    LG_CHECK_LAUNCH();
    return LG_OK;
}
