// The per-element Adam / AdaBelief update shared by the optimizer kernels (optim.hip) and the optimizer launch that
// also exchanges the gradient bucket between the ranks of a node (p2p.hip).  Arithmetic: optim.hip's header.
#pragma once
#include "common.h"

namespace lg {

struct AdamScalars {
    float neg_lr, b1, one_minus_b1, b2, one_minus_b2, eps, inv_bias1, inv_bias2, gscale;
    int   belief, scale_grad;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamScalars& c) {
    if (c.scale_grad) g = g * c.gscale;
    m = c.b1 * m + c.one_minus_b1 * g;
    const float s = c.belief ? g - m : g;
    v = c.b2 * v + c.one_minus_b2 * (s * s);
    const float mh = m * c.inv_bias1, vh = v * c.inv_bias2;
    p = p + (c.neg_lr * mh) * (1.0f / (sqrtf(vh) + c.eps));
}

// the host scalars of one update, rounded once to fp32 like numpy rounds a python float that meets an fp32 array
inline AdamScalars adam_scalars(double lr, double b1, double b2, double eps, double inv_bias1, double inv_bias2, double gscale, int belief) {
    AdamScalars c;
    c.neg_lr = float(-lr); c.b1 = float(b1); c.one_minus_b1 = float(1.0 - b1); c.b2 = float(b2); c.one_minus_b2 = float(1.0 - b2);
    c.eps = float(eps); c.inv_bias1 = float(inv_bias1); c.inv_bias2 = float(inv_bias2); c.gscale = float(gscale); c.belief = belief;
    c.scale_grad = gscale != 1.0;
    return c;
}

// ---- the update applied where the gradient is made (lg_adam_plan_* / lg_adam_epilogue_*, optim.hip) --------------------
// One parameter's update as data in DEVICE memory: the kernel that produces the parameter's gradient (GEMM epilogue of
// gemm_tile_body.inc, the slab workgroups of head.hip) applies it to the values it is about to store, so the optimizer's own
// launch disappears from the step.  The new parameter values go to a SECOND buffer (p_out != p_in): other workgroups of the
// same launch still read the old ones (dx = g @ W next to dW = g^T @ x in sgemm_pair_wgrad_xgrad; the dx tiles next to the
// dW slabs in head_bwd).  The step number behind the bias corrections alternates between two words the same way: every
// workgroup reads step_in, which nobody writes during this step, and exactly one workgroup per step - the one that handles
// element 0 of the plan whose step_out is not NULL - writes step_out = step_in + 1.
struct AdamPlan {
    const float*   p_in;
    float*         p_out;
    float*         m;
    float*         v;
    const int64_t* step_in;     // optimizer steps done so far
    int64_t*       step_out;    // NULL except in the one plan of the optimizer that advances the step number
    int64_t        n;           // elements
    int64_t        t_mul, t_add;   // t = steps_done * t_mul + t_add (the reference advances t once per PARAMETER, optim.py:36/:48)
    double         b1, b2;
    AdamScalars    c;           // inv_bias1 / inv_bias2 are filled in on the device
    // The bias corrections 1/(1 - b^t) of every step this plan will ever see, made on the HOST when the plan is created
    // (libm pow in double, rounded once to fp32: the python expression `(1 - self.b1**self.t)` to the letter): entry s holds the
    // pair for steps_done == s; from entry table_steps - 1 on both are exactly 1.0f (b^t < 2^-25).  Two double-precision pow()
    // by one lane cost a wavefront ~2 us - in an epilogue that sits on the critical path of its launch they cost more than the
    // optimizer launch they replace (measured: the MLP step 59.7 -> 64.7 us with them, bench_v1 of round 4).  NULL: too many
    // steps to tabulate (b2 very close to 1): the powers are formed on the device as before.
    const float*   table;
    int64_t        table_steps;
};

// the scalars of this step for one wavefront: lane 0 reads the step number and forms the two double-precision powers
// (as the python expression does), every lane receives them - no LDS, no workgroup barrier, callable where only some waves
// of a workgroup are still alive
__device__ __forceinline__ AdamScalars adam_plan_scalars(const AdamPlan* pl, int64_t& steps_done) {
    AdamScalars c = pl->c;
    if (pl->table) {                          // uniform: kernel-argument pointer, scalar loads
        steps_done = pl->step_in[0];
        const int64_t s = steps_done < pl->table_steps ? steps_done : pl->table_steps - 1;
        c.inv_bias1 = pl->table[2 * s];
        c.inv_bias2 = pl->table[2 * s + 1];
        return c;
    }
    float i1 = 0.f, i2 = 0.f;
    long long done = 0;
    if ((threadIdx.x & 63) == 0) {
        done = __hip_atomic_load(pl->step_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double t = double(done * pl->t_mul + pl->t_add);
        i1 = float(1.0 / (1.0 - pow(pl->b1, t)));
        i2 = float(1.0 / (1.0 - pow(pl->b2, t)));
    }
    c.inv_bias1 = __shfl(i1, 0, 64);
    c.inv_bias2 = __shfl(i2, 0, 64);
    steps_done = (long long)(__shfl((unsigned long long)done, 0, 64));
    return c;
}

// host side (optim.hip): the plan armed for the gradient buffer `grad` of `n` floats, if the launch being prepared may apply
// it - i.e. it OVERWRITES the gradient (accumulate == 0: the first and only write of this step).  The plan then counts as
// applied.  NULL: nothing armed, or not applicable (the optimizer's lg_adam_epilogue_finish applies what is left).
// A launch that ADDS into a gradient whose plan was already applied in this step is an error: *rc = LG_EINVAL.
const AdamPlan* adam_epilogue_take(const void* grad, int64_t n, int accumulate, int* rc);
// a launch that only adds into / reads gradients: fails when `grad` belongs to a plan already applied in this step
int adam_epilogue_check_write(const void* grad);

}  // namespace lg
