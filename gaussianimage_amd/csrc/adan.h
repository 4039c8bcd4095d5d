// adan.h -- the Adan update of one element and the tensor tables the fused kernels take
// (optimizer.py:296-362 of the reference).
#pragma once
#include "common.h"

#define ADAN_MAX_TENSORS 8

struct AdanTensors {
    float *p[ADAN_MAX_TENSORS];
    float *g[ADAN_MAX_TENSORS];
    float *m[ADAN_MAX_TENSORS];
    float *n[ADAN_MAX_TENSORS];
    float *d[ADAN_MAX_TENSORS];
    float *prev[ADAN_MAX_TENSORS];
    long long numel[ADAN_MAX_TENSORS];
};

struct AdanScalars {
    float beta1, omb1, beta2, omb2, beta3, omb3;
    float bc3_sqrt, eps, neg_step, neg_step_diff, decay_mul, decay_div;
    int no_prox, first_step;
};

// One element of the update, optimizer.py:319-362 in order.
__device__ __forceinline__ void adan_element(const AdanScalars &h, float g, float &p, float &m,
                                             float &n, float &d, float &prev)
{
    const float np0 = h.first_step ? -g : prev;     // optimizer.py:187-189
    const float diff = np0 + g;                     // :323
    m = m * h.beta1 + h.omb1 * g;                   // :325-326
    d = d * h.beta2 + h.omb2 * diff;                // :328-330
    const float u = diff * h.beta2 + g;             // :332-333
    n = n * h.beta3 + h.omb3 * (u * u);             // :334-338
    const float denom = sqrtf(n) / h.bc3_sqrt + h.eps;  // :340-342
    if (h.no_prox) p = p * h.decay_mul;             // :348
    p = p + h.neg_step * (m / denom);               // :349 / :355
    p = p + h.neg_step_diff * (d / denom);          // :350 / :356
    if (!h.no_prox) p = p / h.decay_div;            // :360
    prev = -g;                                      // :361-362
}

__device__ __forceinline__ void adan_inplace(const AdanScalars &h, float g, float *p, float *m,
                                             float *n, float *d, float *prev, size_t i)
{
    float pi = p[i], mi = m[i], ni = n[i], di = d[i], pv = prev[i];
    adan_element(h, g, pi, mi, ni, di, pv);
    p[i] = pi;
    m[i] = mi;
    n[i] = ni;
    d[i] = di;
    prev[i] = pv;
}

// Tail of a fitting iteration, one lane per Gaussian: sum the Gaussian's backward partials,
// chain them through the projection and the model's activations to the raw parameters
// (gaussianimage_cholesky.py:62,74 / gaussianimage_rs.py:55-64), and take the Adan step on its
// 8 (Cholesky) or 8 (RS: 2+2+1+3) parameters -- no gradient tensor is materialised.
// Tensor slots: 0 = _xyz[N,2], 1 = _cholesky[N,3] or _scaling[N,2], 2 = _rotation[N,1] (RS),
// 3 = _features_dc[N,3].
// One parameter element with its optimizer state, held in registers.
struct AdanSlot {
    float p, m, n, d, prev;
};

__device__ __forceinline__ AdanSlot adan_load(const AdanTensors &T, int k, size_t i)
{
    AdanSlot s;
    s.p = T.p[k][i];
    s.m = T.m[k][i];
    s.n = T.n[k][i];
    s.d = T.d[k][i];
    s.prev = T.prev[k][i];
    return s;
}

// Updates the element in memory and returns the new parameter value.
__device__ __forceinline__ float adan_apply(const AdanScalars &h, const AdanTensors &T, int k,
                                            size_t i, AdanSlot s, float g)
{
    adan_element(h, g, s.p, s.m, s.n, s.d, s.prev);
    T.p[k][i] = s.p;
    T.m[k][i] = s.m;
    T.n[k][i] = s.n;
    T.d[k][i] = s.d;
    T.prev[k][i] = s.prev;
    return s.p;
}
