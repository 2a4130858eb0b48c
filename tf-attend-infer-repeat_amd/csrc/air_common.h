// Shared device helpers for libair_hip.so (gfx950 only).
// The translation units are built with -ffp-contract=off: every fp32 op rounds
// once, in source order, so the sampler / loss kernels reproduce the op order of
// the reference graph.  FMAs are written explicitly
// (__builtin_fmaf) where fusion is wanted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "air_hip.h"

#define AIR_EPS 1e-9f   // the sources' 10e-10 (air_model.py:95,587; concrete.py:20,33)

#define AIR_CHECK_LAUNCH()                              \
    do {                                                \
        hipError_t e__ = hipGetLastError();             \
        if (e__ != hipSuccess) return (int)e__;         \
    } while (0)

static inline hipStream_t air_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Opt-in to more than 48 KB of dynamic LDS, ONCE per (kernel function, device): hipFuncSetAttribute is a host-side
// driver call and must not sit on every launch -- in particular not inside stream capture (callers warm up eagerly).
// The attribute is per device, so the cache is one row of kernel functions per device (one table per translation
// unit).  Thread-safe: a function is published into its device's row (compare-and-swap on a free slot) only AFTER the
// driver call returned, so a reader that finds it may launch; two threads racing on the same (function, device) both
// make the call, which is idempotent.  A full row or a device index beyond the table is an error, not a silent
// per-launch driver call.
static inline int air_grant_lds(const void* fn, size_t bytes) {
    constexpr int MAX_DEV = 32, SLOTS = 64;
    static std::atomic<const void*> granted[MAX_DEV][SLOTS];
    if (bytes > 160 * 1024) return AIR_ELIMIT;
    if (bytes <= 48 * 1024) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 0 || dev >= MAX_DEV) return AIR_ELIMIT;
    std::atomic<const void*>* row = granted[dev];
    for (int i = 0; i < SLOTS; ++i) {
        const void* g = row[i].load(std::memory_order_acquire);
        if (g == fn) return 0;
        if (!g) break;                                   // slots fill front to back and are never cleared
    }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    for (int i = 0; i < SLOTS; ++i) {
        const void* expect = nullptr;
        if (row[i].compare_exchange_strong(expect, fn, std::memory_order_acq_rel) || expect == fn) return 0;
    }
    return AIR_ELIMIT;
}

__device__ __forceinline__ float air_sigmoid(float x) {
    // tf.nn.sigmoid: 1 / (1 + exp(-x))
    return 1.0f / (1.0f + expf(-x));
}

__device__ __forceinline__ float air_softplus(float x) {
    // TF 1.3 softplus_op.h: threshold = log(eps)+2 = -13.9424
    const float thr = -13.942384719848633f;
    if (x > -thr) return x;
    if (x < thr) return expf(x);
    return logf(expf(x) + 1.0f);
}

// two fp32 -> packed bf16 pair / one value, round to nearest even (v_cvt_pk_bf16_f32): the rounding every
// bf16-operand kernel applies, so a twin written by a producer is what its consumer would have made itself
typedef __bf16 air_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float air_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned air_pack_bf16(float lo, float hi) {
    const air_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, air_bf16x2_t));
}
__device__ __forceinline__ unsigned short air_bf16_of(float v) { return (unsigned short)(air_pack_bf16(v, 0.0f) & 0xffffu); }

// wave64 butterfly sum: every lane ends with the total
__device__ __forceinline__ float air_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Phase stamps of workgroup (0,0) for tools/phase_stamps.py (debug builds with -DAIR_STAMPS only;
// the shipped library compiles them away).  wall_clock64(): 100 MHz constant counter.
#ifdef AIR_STAMPS
static __device__ unsigned long long air_stamps_dev[64];       // one array per translation unit
// reader of this translation unit's stamps (instantiate once per .hip that places stamps)
#define AIR_STAMPS_READER(fn) extern "C" int fn(unsigned long long* out, int n) { \
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(air_stamps_dev), sizeof(unsigned long long) * (n < 64 ? n : 64)); }
#define AIR_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) air_stamps_dev[i] = wall_clock64(); } while (0)
#else
#define AIR_STAMPS_READER(fn)
#define AIR_STAMP(i) do { } while (0)
#endif

// four block-wide sums at once (one barrier pair) for NW waves; `red` is >= 4*NW floats of LDS.
// Deterministic: fixed butterfly + fixed wave order.
template <int NW>
__device__ __forceinline__ void air_block_sum4(float (&v)[4], float* red) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = air_wave_sum(v[k]);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[k * NW + wave] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float t = red[k * NW];
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) t += red[k * NW + wv];
        v[k] = t;
    }
}

// block-wide sum for NW waves; `red` is >= NW floats of LDS.  Fixed order; all threads get the total.
template <int NW>
__device__ __forceinline__ float air_block_sum_n(float v, float* red) {
    v = air_wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int wv = 1; wv < NW; ++wv) t += red[wv];
    return t;
}

// block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 4 floats of LDS.
// Deterministic: fixed butterfly + fixed wave order.  All threads get the total.
__device__ __forceinline__ float air_block_sum_256(float v, float* red) {
    v = air_wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

// clip_by_global_norm + ApplyAdam coefficients, identical in every workgroup: the partial sums are
// re-reduced in one fixed order (air_model.py:673, TF 1.3 training_ops ApplyAdam).
struct AirAdamCoef { float scale, lr_t, gnorm; };
// in two halves so a streaming kernel can put its first operand loads between them: this thread's share of the partial
// sums (loads only), then the block reduction and the scalar arithmetic
__device__ __forceinline__ float air_adam_partial_share(const float* __restrict__ partials, int npartials) {
    float s = 0.0f;
    for (int i = threadIdx.x; i < npartials; i += 256) s += partials[i];
    return s;
}
__device__ __forceinline__ AirAdamCoef air_adam_coef_from_share(float s, const float* __restrict__ dyn, const int32_t* __restrict__ istate,
                                                                float prescale, float b1, float b2, float* red /* >= 4 floats of LDS */) {
    s = air_block_sum_256(s, red);
    AirAdamCoef c;
    c.gnorm = sqrtf(s) * prescale;                     // norm of the (pre-scaled, e.g. averaged) gradient
    const float clip = dyn[AIR_DYN_CLIP_NORM];
    // t * clip_norm * min(1/global_norm, 1/clip_norm); clip <= 0 disables clipping
    c.scale = prescale * (clip > 0.0f ? clip * fminf(1.0f / c.gnorm, 1.0f / clip) : 1.0f);
    const float t = (float)istate[AIR_IST_GLOBAL_STEP];     // already incremented (grad_sqnorm / fused wgrad)
    c.lr_t = dyn[AIR_DYN_LEARNING_RATE] * sqrtf(1.0f - powf(b2, t)) / (1.0f - powf(b1, t));
    return c;
}
__device__ __forceinline__ AirAdamCoef air_adam_coef(const float* __restrict__ partials, int npartials,
                                                     const float* __restrict__ dyn, const int32_t* __restrict__ istate,
                                                     float prescale, float b1, float b2, float* red /* >= 4 floats of LDS */) {
    return air_adam_coef_from_share(air_adam_partial_share(partials, npartials), dyn, istate, prescale, b1, b2, red);
}
// m += (g-m)(1-b1); v += (g^2-v)(1-b2); var -= lr_t*m/(sqrt(v)+eps)
__device__ __forceinline__ void air_adam_update(float& p, float& m, float& v, float g, const AirAdamCoef& c,
                                                float omb1, float omb2, float eps) {
    const float gk = g * c.scale;
    m = m + (gk - m) * omb1;
    v = v + (gk * gk - v) * omb2;
    p = p - (m * c.lr_t) / (sqrtf(v) + eps);
}
