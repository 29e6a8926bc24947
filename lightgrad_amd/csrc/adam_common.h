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

}  // namespace lg
