// Bit-exact layout kernels: strided copy and strided fill for any item size.
// They back contiguous()/copy()/__getitem__/__setitem__/fill()/zeros()/ones() of HipTensor
// (reference: `atom` with op 'o = a' / 'd = s', opencl/tensor.py:103-116, opencl/ops.py:322-340,
// and clEnqueueFillBuffer, opencl/ops.py:172-177).  4-byte items take the vectorised /
// LDS-transposing paths of the elementwise engine (a register move never changes bits).
#include "common.h"

namespace lg {

template <typename T, typename IdxT>
__global__ void __launch_bounds__(256) copy_gather(T* __restrict__ dst, const T* __restrict__ src, IterDesc d) {
    const int nd = d.ndim;
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < d.numel; e += stride) {
        IdxT rem = IdxT(e);
        int64_t od = 0, os = 0;
        for (int k = nd - 1; k >= 0; --k) {
            IdxT sz = IdxT(d.shape[k]);
            IdxT idx = rem % sz;
            rem /= sz;
            od += int64_t(idx) * d.stride[0][k];
            os += int64_t(idx) * d.stride[1][k];
        }
        dst[od] = src[os];
    }
}

template <typename T, typename IdxT>
__global__ void __launch_bounds__(256) fill_gather(T* __restrict__ dst, IterDesc d, T value) {
    const int nd = d.ndim;
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < d.numel; e += stride) {
        IdxT rem = IdxT(e);
        int64_t od = 0;
        for (int k = nd - 1; k >= 0; --k) {
            IdxT sz = IdxT(d.shape[k]);
            IdxT idx = rem % sz;
            rem /= sz;
            od += int64_t(idx) * d.stride[0][k];
        }
        dst[od] = value;
    }
}

// contiguous fill: 16-byte stores over the aligned body, items over head and tail
template <typename T>
__global__ void __launch_bounds__(256) fill_flat(T* __restrict__ dst, int64_t n, T value, int64_t head, int64_t nvec) {
    constexpr int PER = 16 / sizeof(T);
    union { T t[PER]; uint4 v; } pack;
#pragma unroll
    for (int i = 0; i < PER; ++i) pack.t[i] = value;
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    int64_t tid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    uint4* body = reinterpret_cast<uint4*>(dst + head);
    for (int64_t v = tid; v < nvec; v += stride) body[v] = pack.v;
    // head [0, head) and tail [head + nvec*PER, n): fewer than 2*PER items in total
    int64_t tail0 = head + nvec * PER;
    if (tid < head) dst[tid] = value;
    if (tid < n - tail0) dst[tail0 + tid] = value;
}

template <typename T>
static int copy_typed(void* dst, const void* src, const IterDesc& d) {
    hipStream_t s = rt().stream;
    if (d.numel < (int64_t(1) << 31))
        hipLaunchKernelGGL((copy_gather<T, uint32_t>), dim3(stream_grid(d.numel)), dim3(256), 0, s, static_cast<T*>(dst),
                           static_cast<const T*>(src), d);
    else
        hipLaunchKernelGGL((copy_gather<T, uint64_t>), dim3(stream_grid(d.numel)), dim3(256), 0, s, static_cast<T*>(dst),
                           static_cast<const T*>(src), d);
    return LG_OK;
}

template <typename T>
static int fill_typed(void* dst, const IterDesc& d, uint64_t bits) {
    hipStream_t s = rt().stream;
    T value;
    memcpy(&value, &bits, sizeof(T));
    if (d.ndim == 1 && (d.stride[0][0] == 1 || d.numel == 1)) {
        constexpr int PER = 16 / sizeof(T);
        uintptr_t addr = reinterpret_cast<uintptr_t>(dst);
        int64_t head = ((16 - (addr & 15u)) & 15u) / sizeof(T);
        if (head > d.numel) head = d.numel;
        int64_t nvec = (d.numel - head) / PER;
        int64_t work = nvec > 2 * PER ? nvec : 2 * PER;
        hipLaunchKernelGGL((fill_flat<T>), dim3(stream_grid(work)), dim3(256), 0, s, static_cast<T*>(dst), d.numel, value, head,
                           nvec);
        return LG_OK;
    }
    if (d.numel < (int64_t(1) << 31))
        hipLaunchKernelGGL((fill_gather<T, uint32_t>), dim3(stream_grid(d.numel)), dim3(256), 0, s, static_cast<T*>(dst), d, value);
    else
        hipLaunchKernelGGL((fill_gather<T, uint64_t>), dim3(stream_grid(d.numel)), dim3(256), 0, s, static_cast<T*>(dst), d, value);
    return LG_OK;
}

}  // namespace lg

using namespace lg;

extern "C" int lg_copy_strided(int itemsize, int ndim, const int64_t* shape, void* dst, const int64_t* dst_strides,
                               const void* src, const int64_t* src_strides) {
    LG_REQUIRE_INIT();
    LG_ARG(itemsize == 1 || itemsize == 2 || itemsize == 4 || itemsize == 8, "lg_copy_strided: itemsize %d not in {1,2,4,8}", itemsize);
    LG_ARG(ndim >= 0 && ndim <= LG_MAX_DIMS, "lg_copy_strided: ndim %d out of range [0, %d]", ndim, LG_MAX_DIMS);
    LG_ARG(dst && src, "lg_copy_strided: NULL pointer");
    LG_ARG(ndim == 0 || (shape && dst_strides && src_strides), "lg_copy_strided: NULL shape/strides");
    const int64_t* strides[2] = {dst_strides, src_strides};
    IterDesc d;
    LG_ARG(build_iter(ndim, shape, strides, 2, d), "lg_copy_strided: bad shape");
    if (d.numel == 0) return LG_OK;
    for (int k = 0; k < d.ndim; ++k)
        LG_ARG(d.shape[k] == 1 || d.stride[0][k] != 0, "lg_copy_strided: destination has a zero stride over an extent > 1");
    // both sides one contiguous run: plain device copy
    if (d.ndim == 1 && (d.numel == 1 || (d.stride[0][0] == 1 && d.stride[1][0] == 1)))
        return lg_memcpy_d2d(dst, src, size_t(d.numel) * itemsize);
    if (itemsize == 4)
        return lg_ew(LG_EW_COPY, ndim, shape, dst, dst_strides, nullptr, nullptr, src, src_strides, nullptr, nullptr, nullptr,
                     nullptr, nullptr, nullptr, 0.0f);
    int rc;
    switch (itemsize) {
        case 1: rc = copy_typed<uint8_t>(dst, src, d); break;
        case 2: rc = copy_typed<uint16_t>(dst, src, d); break;
        default: rc = copy_typed<uint64_t>(dst, src, d); break;
    }
    if (rc != LG_OK) return rc;
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_fill_strided(int itemsize, int ndim, const int64_t* shape, void* dst, const int64_t* dst_strides,
                               uint64_t value_bits) {
    LG_REQUIRE_INIT();
    LG_ARG(itemsize == 1 || itemsize == 2 || itemsize == 4 || itemsize == 8, "lg_fill_strided: itemsize %d not in {1,2,4,8}", itemsize);
    LG_ARG(ndim >= 0 && ndim <= LG_MAX_DIMS, "lg_fill_strided: ndim %d out of range [0, %d]", ndim, LG_MAX_DIMS);
    LG_ARG(dst != nullptr, "lg_fill_strided: NULL pointer");
    LG_ARG(ndim == 0 || (shape && dst_strides), "lg_fill_strided: NULL shape/strides");
    const int64_t* strides[1] = {dst_strides};
    IterDesc d;
    LG_ARG(build_iter(ndim, shape, strides, 1, d), "lg_fill_strided: bad shape");
    if (d.numel == 0) return LG_OK;
    int rc;
    switch (itemsize) {
        case 1: rc = fill_typed<uint8_t>(dst, d, value_bits); break;
        case 2: rc = fill_typed<uint16_t>(dst, d, value_bits); break;
        case 4: rc = fill_typed<uint32_t>(dst, d, value_bits); break;
        default: rc = fill_typed<uint64_t>(dst, d, value_bits); break;
    }
    if (rc != LG_OK) return rc;
    LG_CHECK_LAUNCH();
    return LG_OK;
}
