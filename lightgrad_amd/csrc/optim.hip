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
#include <vector>
#include <cmath>

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
// blockIdx.y selects the parameter, whose step number is t = steps_done * nseg + j + 1 (optim.py:36/:48).
//
// Who advances the step number.  With step_slots > 0 the launch does, and without any hand-off between workgroups:
// step[2 + s] is the private copy of the workgroup with slot s = slot_base + blockIdx.y * gridDim.x + blockIdx.x - read
// when it starts, incremented when it ends, by the same workgroup index in every launch (same bucket, same grid) - and
// one workgroup mirrors it into step[0] for readers.  A shared word would need a "last workgroup" (an arrival ticket
// per workgroup: measured 2 % of the MLP step) or a carrier kernel elsewhere in the step (round 2: the loss kernel of
// the NEXT step, with bookkeeping across hipGraph captures that went wrong for separately captured graphs).
namespace lg {

constexpr int kMaxSegments = 64;
struct AdamSegments {
    int     nseg;          // parameters in THIS launch (<= kMaxSegments)
    int     nseg_total;    // parameters of the optimizer: the reference's `t` advances once per PARAMETER (optim.py:36/:48)
    int     first;         // index of this launch's first parameter
    int     slot_base;     // step slot of workgroup 0 of this launch
    int     mirror_slot;   // the workgroup with this slot also writes step[0] (-1: none in this launch)
    // COMPACT grid (round 4): parameter j owns workgroups wg_base[j] .. wg_base[j+1] - one per 1024 elements - instead of a row of
    // a 2-D grid as wide as the LONGEST parameter needs: for the MNIST MLP that grid had 1568 workgroups of which 399 had work,
    // and dispatching the 1169 that return at once is not free (the same lesson as the tail jobs of round 3)
    int     wg_base[kMaxSegments + 1];
    int64_t offsets[kMaxSegments + 1];
};

__global__ void __launch_bounds__(256) adam_multi_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, AdamSegments seg, AdamScalars c,
                                                      int64_t* __restrict__ step, double b1, double b2, int own_slots, int base_aligned) {
    __shared__ float inv_bias[2];
    const int b = blockIdx.x;
    int j = 0;
    while (j + 1 < seg.nseg && b >= seg.wg_base[j + 1]) ++j;          // uniform: scalar loads
    const int local = b - seg.wg_base[j], wgs = seg.wg_base[j + 1] - seg.wg_base[j];
    const int64_t begin = seg.offsets[j], n = seg.offsets[j + 1] - begin;
    // four elements per thread where the segment allows 16-byte accesses (vec == 1), else one
    const int vec = (base_aligned && (begin & 3) == 0) ? 1 : 0;
    const int slot = seg.slot_base + b;
    int64_t steps_done = 0;
    if (threadIdx.x == 0) {
        // the two double-precision powers once per workgroup, not once per thread (they were most of the kernel)
        steps_done = __hip_atomic_load(own_slots ? step + 2 + slot : step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double t = double(steps_done * seg.nseg_total + seg.first + j + 1);
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
    const int64_t stride = int64_t(wgs) * blockDim.x;
    if (vec) {
        const int64_t nvec = n / 4;
        for (int64_t i = int64_t(local) * blockDim.x + threadIdx.x; i < nvec; i += stride) {
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
        if (local == 0)
            for (int64_t i = nvec * 4 + threadIdx.x; i < n; i += blockDim.x) adam_elem(P[i], G[i], M[i], V[i], c);
    } else {
        for (int64_t i = int64_t(local) * blockDim.x + threadIdx.x; i < n; i += stride) adam_elem(P[i], G[i], M[i], V[i], c);
    }
    if (own_slots && threadIdx.x == 0) {
        __hip_atomic_store(step + 2 + slot, steps_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (slot == seg.mirror_slot) __hip_atomic_store(step, steps_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace lg

// launches one group of <= kMaxSegments parameters; returns the number of step slots the group's grid spans in *slots_used
static int lg_adam_multi_group(float* p, const float* g, float* m, float* v, int nseg, const int64_t* offsets, int first, int nseg_total,
                               double lr, double b1, double b2, double eps, int64_t* step, int64_t step_slots, int slot_base, bool* mirrored,
                               double gscale, int belief, int* slots_used) {
    AdamSegments seg;
    seg.nseg = nseg;
    seg.nseg_total = nseg_total;
    seg.first = first;
    seg.slot_base = slot_base;
    seg.mirror_slot = -1;
    int64_t total = 0;
    seg.wg_base[0] = 0;
    for (int j = 0; j <= nseg; ++j) {
        seg.offsets[j] = offsets[j];
        if (j > 0) {
            LG_ARG(offsets[j] >= offsets[j - 1], "lg_adam_multi_dev_f32: offsets must be non-decreasing");
            total += (offsets[j] - offsets[j - 1] + 1023) / 1024;          // one workgroup per 1024 elements (four per thread)
            LG_ARG(total < (int64_t(1) << 22), "lg_adam_multi_dev_f32: bucket too large for one launch");
            seg.wg_base[j] = int(total);
        }
    }
    *slots_used = 0;
    if (total == 0) return LG_OK;
    const AdamScalars c = adam_scalars(lr, b1, b2, eps, 0.0, 0.0, gscale, belief);
    const int base_aligned = (aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v)) ? 1 : 0;
    *slots_used = int(total);
    if (step_slots > 0) {
        LG_ARG(slot_base + *slots_used <= step_slots, "lg_adam_multi_dev_f32: the grid spans %d step slots, the caller gave %lld (lghip.h)",
               slot_base + *slots_used, (long long)step_slots);
        if (!*mirrored)                                   // the first workgroup of the first non-empty parameter keeps step[0] current
            for (int j = 0; j < nseg; ++j)
                if (offsets[j + 1] > offsets[j]) { seg.mirror_slot = slot_base + seg.wg_base[j]; *mirrored = true; break; }
    }
    hipLaunchKernelGGL(adam_multi_dev, dim3(unsigned(total)), dim3(256), 0, rt().stream, p, g, m, v, seg, c, step, b1, b2,
                       step_slots > 0 ? 1 : 0, base_aligned);
    LG_CHECK_LAUNCH();
    return LG_OK;
}

extern "C" int lg_adam_multi_dev_f32(float* p, const float* g, float* m, float* v, int nseg, const int64_t* offsets,
                                     double lr, double b1, double b2, double eps, int64_t* step, int64_t step_slots, double gscale,
                                     int belief) {
    LG_REQUIRE_INIT();
    LG_ARG(nseg >= 1, "lg_adam_multi_dev_f32: %d segments", nseg);
    LG_ARG(p && g && m && v && step && offsets, "lg_adam_multi_dev_f32: NULL pointer");
    LG_ARG(step_slots >= 0, "lg_adam_multi_dev_f32: negative step_slots");
    // more parameters than one launch's argument block holds (tiny-BERT has 40+, larger models hundreds): groups of
    // kMaxSegments, one launch each, every group with step slots of its own
    int slot_base = 0;
    bool mirrored = false;
    for (int first = 0; first < nseg; first += kMaxSegments) {
        const int count = nseg - first < kMaxSegments ? nseg - first : kMaxSegments;
        int used = 0;
        const int rc = lg_adam_multi_group(p, g, m, v, count, offsets + first, first, nseg, lr, b1, b2, eps, step, step_slots, slot_base, &mirrored,
                                           gscale, belief, &used);
        if (rc != LG_OK) return rc;
        slot_base += used;
    }
    return LG_OK;
}

// ---- the update applied where the gradient is made ---------------------------------------------------------------------
// Round 4: the MLP step spent 6 of its 60 us in the optimizer's own launch, which only re-reads what
// the backward kernels have just written.  An optimizer can instead describe every parameter's update once
// (lg_adam_plan_create: two plans per parameter, one per direction between its two value buffers), ARM the plans of the
// coming step for the gradient buffers they belong to (lg_adam_epilogue_arm, in zero_grad) and let the kernels that write
// those gradients apply them (adam_epilogue_take: GEMM epilogues incl. the row-sum column, head_bwd's slab workgroups).
// lg_adam_epilogue_finish applies whatever no kernel took (one launch for all of them; none when every plan was taken)
// and ends the step.  Everything is host bookkeeping + fixed device pointers: a step recorded in a hipGraph replays as is;
// the two directions alternate, so a graph must hold an EVEN number of steps.
namespace lg {

struct ArmedPlan {
    const void*     grad;
    const AdamPlan* plan_dev;
    int64_t         n;
    bool            applied;
};
struct EpilogueState {
    std::vector<ArmedPlan> armed;
};
static EpilogueState& epi() { static EpilogueState s; return s; }

const AdamPlan* adam_epilogue_take(const void* grad, int64_t n, int accumulate, int* rc) {
    if (rc) *rc = LG_OK;
    EpilogueState& E = epi();
    if (E.armed.empty() || grad == nullptr) return nullptr;
    for (ArmedPlan& a : E.armed) {
        if (a.grad != grad) continue;
        if (a.applied) {
            set_error("a kernel writes the gradient at %p again after its optimizer update was applied by the kernel that wrote it "
                      "first in this step (lg_adam_epilogue_arm is for gradients written by ONE kernel per step: no shared weights)", grad);
            if (rc) *rc = LG_EINVAL;
            return nullptr;
        }
        if (accumulate || a.n != n) return nullptr;      // adds to an existing gradient / another extent: left to lg_adam_epilogue_finish
        a.applied = true;
        return a.plan_dev;
    }
    return nullptr;
}

int adam_epilogue_check_write(const void* grad) {
    int rc = LG_OK;
    for (const ArmedPlan& a : epi().armed)
        if (a.grad == grad && a.applied) { (void)adam_epilogue_take(grad, a.n, 1, &rc); break; }
    return rc;
}

constexpr int kPlanBatch = 32;
struct PlanBatch {
    const AdamPlan* plan[kPlanBatch];
    const float*    grad[kPlanBatch];
    int             count;
};

// the plans no kernel took: blockIdx.y = plan, the update of adam_multi_dev with p_in -> p_out
__global__ void __launch_bounds__(256) adam_plan_apply(PlanBatch pb) {
    const AdamPlan* pl = pb.plan[blockIdx.y];
    const float* __restrict__ g = pb.grad[blockIdx.y];
    const int64_t n = pl->n;
    const int64_t first = int64_t(blockIdx.x) * blockDim.x;
    if (first >= n) return;
    int64_t steps_done;
    const AdamScalars c = adam_plan_scalars(pl, steps_done);
    const float* __restrict__ pin = pl->p_in;
    float* __restrict__ pout = pl->p_out;
    float* __restrict__ m = pl->m;
    float* __restrict__ v = pl->v;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = first + threadIdx.x; i < n; i += stride) {
        float P = pin[i], M = m[i], V = v[i];
        adam_elem(P, g[i], M, V, c);
        pout[i] = P; m[i] = M; v[i] = V;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && pl->step_out) __hip_atomic_store(pl->step_out, steps_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace lg

extern "C" int lg_adam_plan_create(void** plan, const float* p_in, float* p_out, float* m, float* v, int64_t n,
                                   const int64_t* step_in, int64_t* step_out, int64_t t_mul, int64_t t_add,
                                   double lr, double b1, double b2, double eps, double gscale, int belief) {
    LG_REQUIRE_INIT();
    LG_ARG(plan && p_in && p_out && m && v && step_in && n >= 1, "lg_adam_plan_create: NULL pointer or empty parameter");
    LG_ARG(p_in != p_out, "lg_adam_plan_create: the new values need a buffer of their own (p_out != p_in): kernels of the same launch still read the old ones");
    LG_ARG(!capturing(), "lg_adam_plan_create: not while capturing a graph (create the plans first)");
    AdamPlan h;
    memset(&h, 0, sizeof(h));
    h.p_in = p_in; h.p_out = p_out; h.m = m; h.v = v; h.step_in = step_in; h.step_out = step_out; h.n = n;
    h.t_mul = t_mul; h.t_add = t_add; h.b1 = b1; h.b2 = b2;
    h.c = adam_scalars(lr, b1, b2, eps, 0.0, 0.0, gscale, belief);
    // the bias corrections of every step, tabulated until both have become exactly 1.0f (header: AdamPlan::table)
    std::vector<float> table;
    constexpr int64_t kMaxTableSteps = int64_t(1) << 21;
    if (b1 >= 0.0 && b1 < 1.0 && b2 >= 0.0 && b2 < 1.0 && t_mul >= 1 && t_add >= 1) {
        for (int64_t s = 0; s < kMaxTableSteps; ++s) {
            const double t = double(s * t_mul + t_add);
            const float i1 = float(1.0 / (1.0 - pow(b1, t))), i2 = float(1.0 / (1.0 - pow(b2, t)));
            table.push_back(i1);
            table.push_back(i2);
            if (i1 == 1.0f && i2 == 1.0f) break;
        }
        if (table[table.size() - 2] != 1.0f || table.back() != 1.0f) table.clear();       // did not converge within the cap
    }
    void* d = nullptr;
    int rc = lg_malloc(&d, sizeof(AdamPlan) + table.size() * sizeof(float));
    if (rc != LG_OK) return rc;
    if (!table.empty()) {
        h.table = reinterpret_cast<const float*>(static_cast<char*>(d) + sizeof(AdamPlan));
        h.table_steps = int64_t(table.size() / 2);
        LG_HIP(hipMemcpyAsync(static_cast<char*>(d) + sizeof(AdamPlan), table.data(), table.size() * sizeof(float), hipMemcpyHostToDevice, rt().stream));
    }
    LG_HIP(hipMemcpyAsync(d, &h, sizeof(h), hipMemcpyHostToDevice, rt().stream));
    LG_HIP(hipStreamSynchronize(rt().stream));          // `h` and `table` live on this stack frame
    *plan = d;
    return LG_OK;
}

extern "C" int lg_adam_plan_destroy(void* plan) {
    if (!plan) return LG_OK;
    for (const ArmedPlan& a : epi().armed)
        LG_ARG(a.plan_dev != plan, "lg_adam_plan_destroy: the plan is armed (lg_adam_epilogue_finish first)");
    return lg_free(plan);
}

extern "C" int lg_adam_epilogue_arm(const float* grad, int64_t n, const void* plan) {
    LG_REQUIRE_INIT();
    LG_ARG(grad && plan && n >= 1, "lg_adam_epilogue_arm: NULL pointer or empty gradient");
    EpilogueState& E = epi();
    for (const ArmedPlan& a : E.armed) LG_ARG(a.grad != grad, "lg_adam_epilogue_arm: a plan is already armed for this gradient (lg_adam_epilogue_finish ends a step)");
    E.armed.push_back(ArmedPlan{grad, static_cast<const AdamPlan*>(plan), n, false});
    return LG_OK;
}

extern "C" int lg_adam_epilogue_disarm(void) {
    epi().armed.clear();
    return LG_OK;
}

extern "C" int lg_adam_epilogue_finish(int* applied_by_kernels, int* applied_here) {
    LG_REQUIRE_INIT();
    EpilogueState& E = epi();
    { const int rc = gemm_group_flush_pending(); if (rc != LG_OK) return rc; }      // queued products write gradients too
    int taken = 0, left = 0;
    PlanBatch pb;
    pb.count = 0;
    auto flush = [&]() {
        if (pb.count == 0) return;
        // (the plans' extents live on the device; the grid is sized for the longest the host knows of)
        int64_t longest = 0;
        for (const ArmedPlan& a : E.armed) if (!a.applied && a.n > longest) longest = a.n;
        hipLaunchKernelGGL(adam_plan_apply, dim3(stream_grid(longest), pb.count), dim3(256), 0, rt().stream, pb);
        pb.count = 0;
    };
    for (const ArmedPlan& a : E.armed) {
        if (a.applied) { ++taken; continue; }
        ++left;
        pb.plan[pb.count] = a.plan_dev;
        pb.grad[pb.count] = static_cast<const float*>(a.grad);
        if (++pb.count == kPlanBatch) flush();
    }
    flush();
    E.armed.clear();
    if (applied_by_kernels) *applied_by_kernels = taken;
    if (applied_here) *applied_here = left;
    LG_CHECK_LAUNCH();
    return LG_OK;
}
