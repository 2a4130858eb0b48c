// Shared pieces of the two sampler translation units (air_sampler.hip: the glimpse read, the heads around it, compose, the
// generic forward; air_sampler_write_bwd.hip: the write backward in its three orders, the generic backward, the lane-order
// probe): the per-axis tap of the axis-aligned transformer, the reference's 4-product expression and the coordinate
// gradient in the saved graph's op order.
#pragma once
#include "air_common.h"
#include <cstdio>
#include <atomic>
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include <cmath>

namespace {

constexpr int THREADS = 256;

struct Tap { float w0, w1; int i0, i1; };   // w0 = (x1_f - x), w1 = (x - x0_f)

// transformer.py:75-87,108-115 for one output coordinate of one axis
__device__ __forceinline__ Tap axis_tap(int j, int n_out, int n_in, float a, float b, float* t_out = nullptr) {
    const float step = 2.0f / (float)(n_out - 1);
    const float t = (n_out > 1) ? (-1.0f + step * (float)j) : -1.0f;   // tf.linspace(-1, 1, n)
    const float xs = a * t + b;                                         // theta . (x_t, y_t, 1)
    const float X = ((xs + 1.0f) * ((float)n_in - 1.001f)) / 2.0f;
    const float f0 = floorf(X);
    const float lim = (float)(n_in - 1);
    const float c0 = fminf(fmaxf(f0, 0.0f), lim);          // clip AFTER floor / +1
    const float c1 = fminf(fmaxf(f0 + 1.0f, 0.0f), lim);
    Tap tp;
    tp.i0 = (int)c0; tp.i1 = (int)c1;
    tp.w0 = c1 - X;
    tp.w1 = X - c0;
    if (t_out) *t_out = t;
    return tp;
}

// literal transformer.py:108-116: wa*Ia + wb*Ib + wc*Ic + wd*Id, add_n left to right
__device__ __forceinline__ float bilinear4(const Tap& tx, const Tap& ty,
                                           float Ia, float Ib, float Ic, float Id) {
    const float wa = tx.w0 * ty.w0;
    const float wb = tx.w0 * ty.w1;
    const float wc = tx.w1 * ty.w0;
    const float wd = tx.w1 * ty.w1;
    return ((wa * Ia + wb * Ib) + wc * Ic) + wd * Id;
}

// Gradient of one output pixel wrt its source coordinates (X, Y) in the op order of the reference's SAVED graph
// (model/air-model.meta, executed by the graph executor of tests/test_graph_exec.py).  For an out-of-range pixel (both
// taps clipped to one index) the four legs cancel exactly in real arithmetic but NOT in fp32: the rounding residue,
// multiplied by g ~ 1 / (r + 1e-9) at unexplained ink, is not noise to be cleaned up -- it is the force that pulls glimpses
// towards unexplained ink, and the reference's training dynamics depend on it (with the exact adjoint the model does not
// learn to localise).  cx = (n_in - 1.001): x = (x_s + 1) * cx / 2.  d wa..wd = g*Ia..Id (mul_10..13_grad), each product's two factors get
// grad*other (mul_6..9_grad), the Sub nodes negate the (x1-x)/(y1-y) legs, and the four legs that
// reach x (y) are summed by AddN_10 / AddN_20 (AddN_11 / AddN_21) left to right in the order
// wa, wb, wc, wd.  Then x = (x_s + 1)*(W - 1.001)/2: truediv_grad then mul_grad.
__device__ __forceinline__ void graph_dxy(float g, float Ia, float Ib, float Ic, float Id,
                                          const Tap& tx, const Tap& ty, float cx, float& dxs, float& dys) {
    const float ga = g * Ia, gb = g * Ib, gc = g * Ic, gd = g * Id;
    const float dX = ((-(ga * ty.w0) + -(gb * ty.w1)) + gc * ty.w0) + gd * ty.w1;
    const float dY = ((-(tx.w0 * ga) + tx.w0 * gb) + -(tx.w1 * gc)) + tx.w1 * gd;
    dxs = (dX / 2.0f) * cx;
    dys = (dY / 2.0f) * cx;
}

template <typename K>
int ensure_lds(K kernel, size_t bytes) { return air_grant_lds(reinterpret_cast<const void*>(kernel), bytes); }

}  // namespace
