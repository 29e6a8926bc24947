// Fused Adam / AdaBelief update: one pass over p, g, m, v instead of the
// ~14 elementwise launches of the tape form (reference optim.py:36-40, :48-52).  HBM-bound:
// 4 reads + 3 writes of 4 B = 28 B per element.
//
// The arithmetic is the reference's expression sequence, evaluated per element with one rounding
// per operation (the library is built with -ffp-contract=off so nothing is fused):
//   m  = b1*m + (1-b1)*g          v = b2*v + (1-b2)*s^2   (s = g for Adam, g - m for AdaBelief)
//   p += ((-lr) * (m * c1)) * (1 / ((v * c2)^0.5 + eps))   with c1 = 1/(1-b1^t), c2 = 1/(1-b2^t)
// (the tape's `/` is `a * b**-1`, autograd/ops.py:30-36, hence the reciprocal-then-multiply form)
#include "common.h"
#include "adam_common.h"

namespace lg {

__global__ void __launch_bounds__(256) adam_vec4(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, int64_t nvec, AdamScalars c) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<const float4*>(g)[i];
        float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
        adam_elem(P.x, G.x, M.x, V.x, c);
        adam_elem(P.y, G.y, M.y, V.y, c);
        adam_elem(P.z, G.z, M.z, V.z, c);
        adam_elem(P.w, G.w, M.w, V.w, c);
        reinterpret_cast<float4*>(p)[i] = P;
        reinterpret_cast<float4*>(m)[i] = M;
        reinterpret_cast<float4*>(v)[i] = V;
    }
}

__global__ void __launch_bounds__(256) adam_scalar(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t begin, int64_t n, AdamScalars c) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = begin + int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
        adam_elem(p[i], g[i], m[i], v[i], c);
}

}  // namespace lg

using namespace lg;

extern "C" int lg_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1, double b2,
                                double eps, double inv_bias1, double inv_bias2, double gscale, int belief) {
    LG_REQUIRE_INIT();
    LG_ARG(n >= 0, "lg_adam_step_f32: negative length");
    if (n == 0) return LG_OK;
    LG_ARG(p && g && m && v, "lg_adam_step_f32: NULL pointer");
    AdamScalars c;
    // 1-b1 and 1-b2 are formed in double like the python expression `(1 - self.b1)`, then rounded once
    // (python float scalars meet fp32 tensors: numpy rounds the double ONCE to fp32 - same here)
    c.neg_lr = float(-lr); c.b1 = float(b1); c.one_minus_b1 = float(1.0 - b1); c.b2 = float(b2); c.one_minus_b2 = float(1.0 - b2);
    c.eps = float(eps); c.inv_bias1 = float(inv_bias1); c.inv_bias2 = float(inv_bias2); c.gscale = float(gscale); c.belief = belief;
    c.scale_grad = gscale != 1.0;
    hipStream_t s = rt().stream;
    const bool vec = aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v);
    const int64_t nvec = vec ? n / 4 : 0;
    if (nvec > 0) hipLaunchKernelGGL(adam_vec4, dim3(stream_grid(nvec)), dim3(256), 0, s, p, g, m, v, nvec, c);
    if (nvec * 4 < n)
        hipLaunchKernelGGL(adam_scalar, dim3(stream_grid(n - nvec * 4)), dim3(256), 0, s, p, g, m, v, nvec * 4, n, c);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

// ---- graph-safe variant: the step number lives in device memory ---------------------------------
// A captured hipGraph replays fixed kernel arguments, so the bias corrections 1/(1 - b^t) cannot be
// host scalars.  Here t = *step * t_mul + t_add is read on the device (the reference advances `t`
// once per PARAMETER, optim.py:36/:48: parameter i of P at optimizer step s has t = s*P + i + 1) and
// the corrections are formed in double like the python expression, then rounded once to fp32.
namespace lg {

__global__ void __launch_bounds__(256) adam_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                float* __restrict__ v, int64_t n, AdamScalars c, const int64_t* __restrict__ step,
                                                int64_t t_mul, int64_t t_add, double b1, double b2) {
    __shared__ float inv_bias[2];
    if (threadIdx.x == 0) {          // the double-precision powers once per workgroup, not once per thread
        const double t = double(step[0] * t_mul + t_add);
        inv_bias[0] = float(1.0 / (1.0 - pow(b1, t)));
        inv_bias[1] = float(1.0 / (1.0 - pow(b2, t)));
    }
    __syncthreads();
    c.inv_bias1 = inv_bias[0];
    c.inv_bias2 = inv_bias[1];
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) adam_elem(p[i], g[i], m[i], v[i], c);
}

__global__ void counter_add(int64_t* counter, int64_t delta) {
    if (threadIdx.x == 0 && blockIdx.x == 0) counter[0] += delta;
}

}  // namespace lg

extern "C" int lg_adam_step_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1, double b2,
                                    double eps, const int64_t* step, int64_t t_mul, int64_t t_add, double gscale, int belief) {
    LG_REQUIRE_INIT();
    LG_ARG(n >= 0, "lg_adam_step_dev_f32: negative length");
    if (n == 0) return LG_OK;
    LG_ARG(p && g && m && v && step, "lg_adam_step_dev_f32: NULL pointer");
    AdamScalars c;
    c.neg_lr = float(-lr); c.b1 = float(b1); c.one_minus_b1 = float(1.0 - b1); c.b2 = float(b2); c.one_minus_b2 = float(1.0 - b2);
    c.eps = float(eps); c.inv_bias1 = 0.f; c.inv_bias2 = 0.f; c.gscale = float(gscale); c.belief = belief;
    c.scale_grad = gscale != 1.0;
    hipLaunchKernelGGL(adam_dev, dim3(stream_grid(n)), dim3(256), 0, rt().stream, p, g, m, v, n, c, step, t_mul, t_add, b1, b2);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_counter_add_i64(int64_t* counter, int64_t delta) {
    LG_REQUIRE_INIT();
    LG_ARG(counter != nullptr, "lg_counter_add_i64: NULL pointer");
    hipLaunchKernelGGL(counter_add, dim3(1), dim3(64), 0, rt().stream, counter, delta);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

// ---- multi-tensor form: all parameters of a model in ONE launch --------------------------------
// p, g, m, v are flat buckets holding nseg parameters back to back (offsets[j] .. offsets[j+1]);
// blockIdx.y selects the parameter, whose step number is t = *step * nseg + j + 1 (optim.py:36/:48).
namespace lg {

constexpr int kMaxSegments = 64;
struct AdamSegments {
    int     nseg;          // parameters in THIS launch (<= kMaxSegments)
    int     nseg_total;    // parameters of the optimizer: the reference's `t` advances once per parameter (optim.py:36/:48)
    int     first;         // index of this launch's first parameter
    int64_t offsets[kMaxSegments + 1];
};

__global__ void __launch_bounds__(256) adam_multi_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, AdamSegments seg, AdamScalars c,
                                                      int64_t* __restrict__ step, double b1, double b2, int advance, int base_aligned) {
    __shared__ float inv_bias[2];
    const int j = blockIdx.y;
    const int64_t begin = seg.offsets[j], n = seg.offsets[j + 1] - begin;
    // four elements per thread where the segment allows 16-byte accesses (vec == 1), else one
    const int vec = (base_aligned && (begin & 3) == 0) ? 1 : 0;
    const int64_t first = (int64_t(blockIdx.x) * blockDim.x) * (vec ? 4 : 1);
    if (first < n) {                                   // workgroup-uniform
        if (threadIdx.x == 0) {
            // the two double-precision powers once per workgroup, not once per thread (they were most of the kernel)
            const double t = double(step[0] * seg.nseg_total + seg.first + j + 1);
            inv_bias[0] = float(1.0 / (1.0 - pow(b1, t)));
            inv_bias[1] = float(1.0 / (1.0 - pow(b2, t)));
        }
        __syncthreads();
        c.inv_bias1 = inv_bias[0];
        c.inv_bias2 = inv_bias[1];
        float* P = p + begin;
        const float* G = g + begin;
        float* M = m + begin;
        float* V = v + begin;
        if (vec) {
            const int64_t nvec = n / 4, stride = int64_t(gridDim.x) * blockDim.x;
            for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nvec; i += stride) {
                float4 pp = reinterpret_cast<float4*>(P)[i], gg = reinterpret_cast<const float4*>(G)[i];
                float4 mm = reinterpret_cast<float4*>(M)[i], vv = reinterpret_cast<float4*>(V)[i];
                adam_elem(pp.x, gg.x, mm.x, vv.x, c);
                adam_elem(pp.y, gg.y, mm.y, vv.y, c);
                adam_elem(pp.z, gg.z, mm.z, vv.z, c);
                adam_elem(pp.w, gg.w, mm.w, vv.w, c);
                reinterpret_cast<float4*>(P)[i] = pp;
                reinterpret_cast<float4*>(M)[i] = mm;
                reinterpret_cast<float4*>(V)[i] = vv;
            }
            if (blockIdx.x == 0)
                for (int64_t i = nvec * 4 + threadIdx.x; i < n; i += blockDim.x) adam_elem(P[i], G[i], M[i], V[i], c);
        } else {
            const int64_t stride = int64_t(gridDim.x) * blockDim.x;
            for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) adam_elem(P[i], G[i], M[i], V[i], c);
        }
    }
    if (advance > 0 && first < n) {
        // the LAST working workgroup to finish advances the step number (step[1] is an arrival ticket, zero between
        // launches; `advance` = number of workgroups that have work): every workgroup has read step[0] before it draws
        // its ticket, so the write cannot race with a read
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long ticket = atomicAdd(reinterpret_cast<unsigned long long*>(step + 1), 1ULL);
            if (ticket == (unsigned long long)advance - 1) {
                step[1] = 0;
                step[0] = step[0] + 1;
            }
        }
    }
}

}  // namespace lg

static int lg_adam_multi_group(float* p, const float* g, float* m, float* v, int nseg, const int64_t* offsets, int first, int nseg_total,
                               double lr, double b1, double b2, double eps, int64_t* step, double gscale, int belief, int advance);

extern "C" int lg_adam_multi_dev_f32(float* p, const float* g, float* m, float* v, int nseg, const int64_t* offsets,
                                     double lr, double b1, double b2, double eps, int64_t* step, double gscale,
                                     int belief, int advance) {
    LG_REQUIRE_INIT();
    LG_ARG(nseg >= 1, "lg_adam_multi_dev_f32: %d segments", nseg);
    LG_ARG(p && g && m && v && step && offsets, "lg_adam_multi_dev_f32: NULL pointer");
    if (nseg > kMaxSegments) {
        // more parameters than one launch's argument block holds (tiny-BERT has 40+, larger models hundreds): groups of
        // kMaxSegments, one launch each.  Every group reads the same step[0]; with the ticket form the LAST group alone
        // advances it (stream order: the earlier groups have finished reading by then).
        for (int first = 0; first < nseg; first += kMaxSegments) {
            const int count = nseg - first < kMaxSegments ? nseg - first : kMaxSegments;
            const int rc = lg_adam_multi_group(p, g, m, v, count, offsets + first, first, nseg, lr, b1, b2, eps, step, gscale, belief,
                                               (advance && first + count == nseg) ? 1 : 0);
            if (rc != LG_OK) return rc;
        }
        return LG_OK;
    }
    return lg_adam_multi_group(p, g, m, v, nseg, offsets, 0, nseg, lr, b1, b2, eps, step, gscale, belief, advance);
}

static int lg_adam_multi_group(float* p, const float* g, float* m, float* v, int nseg, const int64_t* offsets, int first, int nseg_total,
                               double lr, double b1, double b2, double eps, int64_t* step, double gscale, int belief, int advance) {
    AdamSegments seg;
    seg.nseg = nseg;
    seg.nseg_total = nseg_total;
    seg.first = first;
    int64_t longest = 0;
    for (int j = 0; j <= nseg; ++j) {
        seg.offsets[j] = offsets[j];
        if (j > 0) {
            LG_ARG(offsets[j] >= offsets[j - 1], "lg_adam_multi_dev_f32: offsets must be non-decreasing");
            if (offsets[j] - offsets[j - 1] > longest) longest = offsets[j] - offsets[j - 1];
        }
    }
    if (longest == 0) return LG_OK;
    AdamScalars c;
    c.neg_lr = float(-lr); c.b1 = float(b1); c.one_minus_b1 = float(1.0 - b1); c.b2 = float(b2); c.one_minus_b2 = float(1.0 - b2);
    c.eps = float(eps); c.inv_bias1 = 0.f; c.inv_bias2 = 0.f; c.gscale = float(gscale); c.belief = belief;
    c.scale_grad = gscale != 1.0;
    // sized for four elements per thread (segments that do not start on a 16-byte boundary loop: grid-stride)
    const int base_aligned = (aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v)) ? 1 : 0;
    const int64_t grid_x = stream_grid((longest + 3) / 4);
    if (advance) {                    // number of workgroups with work: they take the arrival tickets
        int64_t active = 0;
        for (int j = 0; j < nseg; ++j) {
            const int64_t n = offsets[j + 1] - offsets[j];
            const int64_t per_wg = 256 * ((base_aligned && (offsets[j] & 3) == 0) ? 4 : 1);
            const int64_t wgs = (n + per_wg - 1) / per_wg;
            active += wgs < grid_x ? wgs : grid_x;
        }
        advance = int(active);
    }
    hipLaunchKernelGGL(adam_multi_dev, dim3(unsigned(grid_x), nseg), dim3(256), 0, rt().stream, p, g, m, v, seg, c, step, b1, b2,
                       advance, base_aligned);
    LG_CHECK_LAUNCH();
    return LG_OK;
}
