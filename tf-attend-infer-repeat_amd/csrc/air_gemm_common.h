// Shared pieces of the GEMM translation units (air_gemm.hip: fp32-operand kernels; air_gemm_bf16.hip:
// bf16-twin-operand kernels): kernel arguments, the prefetched fused epilogues, the cross-wave reduction
// and the XCD-aware tile map.  Everything here is inline device code in namespace airg.
#pragma once
#include "air_common.h"
#include "air_philox.h"
#include <type_traits>

namespace airg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int THREADS = 256;
// internal epilogue id (never in the ABI): AIR_EPI_LSTM_FWD on 16-column tiles of FOUR units x four gates (column
// c = gate c >> 2 of unit n0 + (c & 3)) -- four times as many, four times lighter workgroups than the 64-column
// grouped tiles; same accumulation per element, same epilogue arithmetic (air_gemm_bf16.hip)
constexpr int EPI_LSTM_FWD_Q = 100;

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// one value, the same rounding: what a consumer's staging conversion would have produced from the fp32 array
__device__ __forceinline__ unsigned short bf16_of(float v) { return (unsigned short)(pack_bf16(v, 0.0f) & 0xffffu); }

struct Args {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    int gstride, gwidth;          // grouped column tiles: col(j, c) = n0 + j*gstride + c, valid if n0 + c < gwidth
    int kslab; long slab_stride;  // grid.z split-K: k-range per z, output slab stride (floats)
    const float* bias; const float* addend; const float* aux;
    int ldadd, ldaux, add_slabs; long add_slab_stride;
    float aux_scale;
    int act, actgrad, accumulate, epi;
    // fused-epilogue operands
    const float* p0; const float* p1; const float* p2; const float* p3; const float* p4;
    float* q0; float* q1; float* q2; float* q3;
    int i0, i1;
    // bf16 twins (RNE, same leading dimensions as their fp32 arrays; any may be null).  A16 / B16: the operands as
    // the bf16-operand kernels read them (air_gemm_bf16.hip); C16 / q0_16 / q2_16: twins this launch writes next
    // to C / q0 / q2 for the GEMMs that consume its output.
    const unsigned short* A16; const unsigned short* B16;
    unsigned short* C16; unsigned short* q0_16; unsigned short* q2_16;
    // panel-blocked twin of an untransposed B (air_gemm_t.B16p): element (k, n) at (n / 16) * K * 16 + k * 16 + n % 16;
    // gate-interleaved panels (unit quad p: [k][gate][4 units]) for the four-unit LSTM tiles
    const unsigned short* B16p;
    // optional step prologue (schedules + Philox noise) carried by the workgroups of an extra grid.z
    // plane: the hoisted x.Wx launch needs neither, so the prologue costs no launch of its own
    int job_on; AirStepJob job;   // job_on = number of grid.z planes given to the prologue (0 = none)
};

// Epilogue operands (bias / addend / aux / LSTM state) are PREFETCHED into registers at kernel
// entry, in the lane->element map the epilogue will use: their memory round trip (~2 us when
// the producer ran on another XCD) overlaps the operand panels' instead of following the MFMAs.
// Items of the epilogue: (row-tile i, column-tile j, q) for the generic one, (i, q) for the fused
// ones; item `it` belongs to wave it & 3.  C/D map: row = (lane>>4)*4 + q, col = lane&15.
template <int TM, int TN>
struct Pre {
    static constexpr int NG = TM * TN;            // generic items per wave
    float bias[NG], add[NG], aux[NG];
    float f[40];                                  // fused-epilogue operands (LSTM: 32 slab values + 4 bias + state)
};

// EPI >= 0: the epilogue is a COMPILE-TIME choice of the lean kernels; EPI < 0: a.epi at run time (fallback
// kernels).  With a run-time choice every branch's loads meet at a control-flow join that needs their values
// -- an s_waitcnt vmcnt(0) right behind them, i.e. a memory round trip of its own at kernel entry instead of
// one that overlaps the operand panels'.  For the same reason optional operands are fetched through a
// pointer SELECT (a.A stands in for an absent operand: always a valid address), never through a branch.
template <int TM, int TN, int EPI>
__device__ __forceinline__ void epilogue_prefetch(const Args& a, Pre<TM, TN>& pre, int m0, int n0, int lane, int wave) {
    const int E = EPI < 0 ? a.epi : EPI;
    const float* const safe = a.A;
    if (E == AIR_EPI_GENERIC) {
#pragma unroll
        for (int k = 0; k < TM * TN; ++k) {
            const int it = wave + 4 * k;
            const int i = it / (TN * 4), j = (it >> 2) % TN, q = it & 3;
            const int m = m0 + i * 16 + (lane >> 4) * 4 + q;
            const int n = n0 + j * 16 + (lane & 15);
            const bool ok = m < a.M && n < a.N;
            const bool hb = ok && a.bias, ha = ok && a.addend && a.add_slabs == 1, hx = ok && a.aux;
            // raw values: the epilogue only reads an operand that exists (a select here would need the
            // loaded value at once, i.e. a wait right behind the load)
            pre.bias[k] = *(hb ? a.bias + n : safe);
            pre.add[k] = *(ha ? a.addend + (size_t)m * a.ldadd + n : safe);
            pre.aux[k] = *(hx ? a.aux + (size_t)m * a.ldaux + n : safe);
        }
        return;
    }
    const int q = wave;                            // fused epilogues run with TM == 1: item = q
    const int m = m0 + (lane >> 4) * 4 + q;
    const int u = n0 + (lane & 15);
    const bool ok = (m < a.M) && (u < a.gwidth);
    // p[off] when `have` (else an unrelated valid word: the epilogue applies the same `have` before use) --
    // one unconditional load from a selected address, no select on the loaded value here
    auto fetch = [&](bool have, const float* p, size_t off) __attribute__((always_inline)) { return *(have ? p + off : safe); };
    if (E == AIR_EPI_LSTM_FWD) {
        const int R = a.gwidth;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = u + j * R;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                pre.f[j * 8 + k] = fetch(ok && k < a.add_slabs, a.addend, k * a.add_slab_stride + (size_t)m * a.ldadd + n);
            pre.f[32 + j] = fetch(ok && a.bias, a.bias, n);
        }
        pre.f[36] = fetch(ok, a.p0, (size_t)m * R + u);
    } else if (E == EPI_LSTM_FWD_Q) {
        // item of lane L of wave 0: row L >> 2, unit n0 + (L & 3) -- the operands AIR_EPI_LSTM_FWD fetches, same slots
        // only wave 0 owns items, and only the slabs that exist are fetched: both conditions are wave-uniform, so they
        // are real branches around the loads (no dummy loads, no address arithmetic on the other three waves -- the
        // in-kernel stamps showed 1.5 us of this 3 us kernel going into issuing 37 selected loads per lane on all waves)
        const int R = a.gwidth, mm = m0 + (lane >> 2), uu = n0 + (lane & 3);
        const bool okq = mm < a.M && uu < R;
#pragma unroll
        for (int i = 0; i < 37; ++i) pre.f[i] = 0.0f;
        if (wave == 0) {
            const size_t base = (size_t)(okq ? mm : 0) * a.ldadd + (okq ? uu : 0);
            if (a.add_slabs == 4) {                        // the train step's slab count: 16 loads back to back, no join between them
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pre.f[j * 8 + k] = a.addend[k * a.add_slab_stride + base + (size_t)j * R];
            } else if (a.add_slabs == 1) {                 // x.Wx left whole by the launch that also ran the first step
#pragma unroll
                for (int j = 0; j < 4; ++j) pre.f[j * 8] = a.addend[base + (size_t)j * R];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        pre.f[j * 8 + k] = fetch(k < a.add_slabs, a.addend, k * a.add_slab_stride + base + (size_t)j * R);
            }
            if (a.bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) pre.f[32 + j] = a.bias[(okq ? uu : 0) + j * R];
            }
            pre.f[36] = a.p0[(size_t)(okq ? mm : 0) * R + (okq ? uu : 0)];
        }
    } else if (E == AIR_EPI_LSTM_FWD0) {
        // item of lane L (wave 0): row L >> 2, unit n0 + (L & 3); only the bias is needed (zero state, no addend)
        const int R = a.gwidth, uu = n0 + (lane & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) pre.f[j] = fetch(a.bias && uu < R, a.bias, j * R + uu);
    } else if (E == AIR_EPI_REPARAM_FWD) {
        const int Z = a.gwidth;
        pre.f[0] = fetch(ok && a.bias, a.bias, u);
        pre.f[1] = fetch(ok && a.bias, a.bias, Z + u);
        pre.f[2] = fetch(ok, a.p0, (size_t)m * Z + u);
    } else if (E == AIR_EPI_REPARAM_BWD) {
        const int Z = a.gwidth;
        pre.f[0] = fetch(ok, a.p0, (size_t)m * 2 * Z + u);
        pre.f[1] = fetch(ok, a.p0, (size_t)m * 2 * Z + Z + u);
        pre.f[2] = fetch(ok, a.p1, (size_t)m * Z + u);
        pre.f[3] = fetch(ok, a.p2, (size_t)m * AIR_ATT_STRIDE + AIR_ATT_MASK);
        pre.f[4] = a.p3[AIR_DYN_GRAD_SCALE]; pre.f[5] = a.p3[AIR_DYN_VAE_PV]; pre.f[6] = a.p3[AIR_DYN_VAE_PM];
    } else if (E == AIR_EPI_LSTM_BWD || E == AIR_EPI_LSTM_BWD_TAIL) {
        const int R = a.gwidth;
        const int mm = m - (E == AIR_EPI_LSTM_BWD_TAIL ? a.i0 : 0);          // row within the step's arrays
        const bool okk = ok && mm >= 0;
        const size_t idx = (size_t)(okk ? mm : 0) * R + u;
#pragma unroll
        for (int j = 0; j < 4; ++j) pre.f[j] = fetch(okk, a.p0, (size_t)(okk ? mm : 0) * 4 * R + j * R + u);
        pre.f[4] = fetch(okk, a.p1, idx);
        pre.f[5] = fetch(okk, a.p2, idx);
        pre.f[6] = fetch(okk && a.p3, a.p3, idx);
        pre.f[7] = fetch(okk && a.addend, a.addend, (size_t)m * a.ldadd + u);
    }
}

// Epilogue over the reduced tile values held in Red[(t*4 + q)*64 + lane] (t = i*TN + j).
template <int TM, int TN, int EPI>
__device__ __forceinline__ void epilogue(const Args& a, const Pre<TM, TN>& pre, const float* Red,
                                         int m0, int n0, int lane, int wave) {
    const int E = EPI < 0 ? a.epi : EPI;
    if (E == AIR_EPI_GENERIC) {
#pragma unroll
        for (int k = 0; k < TM * TN; ++k) {
            const int it = wave + 4 * k;
            const int i = it / (TN * 4), j = (it >> 2) % TN, q = it & 3;
            const int m = m0 + i * 16 + (lane >> 4) * 4 + q;
            const int n = n0 + j * 16 + (lane & 15);
            if (m >= a.M || n >= a.N) continue;
            float v = Red[((i * TN + j) * 4 + q) * 64 + lane];
            if (a.bias) v += pre.bias[k];
            if (a.addend) {
                if (a.add_slabs == 1) v += pre.add[k];
                else for (int sl = 0; sl < a.add_slabs; ++sl) v += a.addend[sl * a.add_slab_stride + (size_t)m * a.ldadd + n];
            }
            if (a.act == AIR_ACT_RELU) v = fmaxf(v, 0.0f);
            else if (a.act == AIR_ACT_SOFTPLUS) v = air_softplus(v);
            else if (a.act == AIR_ACT_SIGMOID_NOISE) v = air_sigmoid(v + pre.aux[k] * a.aux_scale);
            if (a.actgrad == AIR_GRAD_RELU) v = (pre.aux[k] > 0.0f) ? v : 0.0f;
            else if (a.actgrad == AIR_GRAD_SOFTPLUS) v = v * (1.0f - expf(-pre.aux[k]));
            float* c = a.C + (size_t)m * a.ldc + n;
            if (a.accumulate) v += *c;
            *c = v;
            if (a.C16) a.C16[(size_t)m * a.ldc + n] = bf16_of(v);
        }
        return;
    }
    if (E == EPI_LSTM_FWD_Q) {
        // tile columns: gate (col >> 2) of unit n0 + (col & 3); wave 0 takes the 64 (row, unit) items.  BasicLSTMCell
        // (air_model.py:286) exactly as AIR_EPI_LSTM_FWD computes it: acc + slabs (in slab order) + bias -> i, j, f, o
        if (wave == 0) {
            const int R = a.gwidth, r = lane >> 2, m = m0 + r, u = n0 + (lane & 3);
            if (m < a.M && u < R) {
                float g[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float s = Red[(r & 3) * 64 + (r >> 2) * 16 + j * 4 + (lane & 3)];
#pragma unroll
                    for (int k = 0; k < 8; ++k) s += (k < a.add_slabs) ? pre.f[j * 8 + k] : 0.0f;   // fixed summation order
                    if (a.bias) s += pre.f[32 + j];
                    g[j] = s;
                }
                const float si = air_sigmoid(g[0]), tj = tanhf(g[1]);
                const float sf = air_sigmoid(g[2] + 1.0f), so = air_sigmoid(g[3]);
                const float cn = pre.f[36] * sf + si * tj;
                float* ac = a.q0 + (size_t)m * 4 * R;
                ac[u] = si; ac[R + u] = tj; ac[2 * R + u] = sf; ac[3 * R + u] = so;
                a.q1[(size_t)m * R + u] = cn;
                const float hn = tanhf(cn) * so;
                a.q2[(size_t)m * R + u] = hn;
                if (a.q2_16) a.q2_16[(size_t)m * R + u] = bf16_of(hn);
            }
        }
        return;
    }
    if (E == AIR_EPI_LSTM_FWD0) {
        // tile columns: gate (col >> 2) of unit n0 + (col & 3).  Every wave stores one accumulator row set of the
        // raw x.Wx; wave 0 then takes the 64 (row, unit) items: BasicLSTMCell from zero state (air_model.py:286, :540)
        const int R = a.gwidth;
        {
            const int m = m0 + (lane >> 4) * 4 + wave, col = lane & 15, u = n0 + (col & 3);
            if (m < a.M && u < R) a.C[(size_t)m * a.ldc + (col >> 2) * R + u] = Red[wave * 64 + lane];
        }
        if (wave == 0) {
            const int r = lane >> 2, m = m0 + r, u = n0 + (lane & 3);
            if (m < a.M && u < R) {
                float g[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float s = Red[(r & 3) * 64 + (r >> 2) * 16 + j * 4 + (lane & 3)];
                    if (a.bias) s += pre.f[j];
                    g[j] = s;
                }
                const float si = air_sigmoid(g[0]), tj = tanhf(g[1]);
                const float sf = air_sigmoid(g[2] + 1.0f), so = air_sigmoid(g[3]);
                const float cn = 0.0f * sf + si * tj;
                float* ac = a.q0 + (size_t)m * 4 * R;
                ac[u] = si; ac[R + u] = tj; ac[2 * R + u] = sf; ac[3 * R + u] = so;
                a.q1[(size_t)m * R + u] = cn;
                const float hn = tanhf(cn) * so;
                a.q2[(size_t)m * R + u] = hn;
                if (a.q2_16) a.q2_16[(size_t)m * R + u] = bf16_of(hn);
            }
        }
        return;
    }
    // fused epilogues: one item = (row-tile i, q); all TN group values of a unit sit in the same lane
    for (int it = wave; it < TM * 4; it += 4) {
        const int i = it >> 2, q = it & 3;
        const int m = m0 + i * 16 + (lane >> 4) * 4 + q;
        const int u = n0 + (lane & 15);
        if (m >= a.M || u >= a.gwidth) continue;
        float v[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) v[j] = Red[((i * TN + j) * 4 + q) * 64 + lane];
        if (E == AIR_EPI_LSTM_FWD) {
            // BasicLSTMCell (air_model.py:286): gates = [x,h].K + b -> i, j, f, o; forget bias 1.0
            // p0 = c_prev [M,R]; addend slabs = hoisted x.Wx; q0 = acts [M,4R], q1 = c, q2 = h
            if (TN == 4 && TM == 1) {
                const int R = a.gwidth;
                float g[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float s = v[j % TN];
#pragma unroll
                    for (int k = 0; k < 8; ++k) s += (k < a.add_slabs) ? pre.f[j * 8 + k] : 0.0f;   // fixed summation order
                    if (a.bias) s += pre.f[32 + j];
                    g[j] = s;
                }
                const float si = air_sigmoid(g[0]), tj = tanhf(g[1]);
                const float sf = air_sigmoid(g[2] + 1.0f), so = air_sigmoid(g[3]);
                const float cn = pre.f[36] * sf + si * tj;
                float* ac = a.q0 + (size_t)m * 4 * R;
                ac[u] = si; ac[R + u] = tj; ac[2 * R + u] = sf; ac[3 * R + u] = so;
                a.q1[(size_t)m * R + u] = cn;
                const float hn = tanhf(cn) * so;
                a.q2[(size_t)m * R + u] = hn;
                if (a.q2_16) a.q2_16[(size_t)m * R + u] = bf16_of(hn);
            }
        } else if (E == AIR_EPI_REPARAM_FWD) {
            // vae.py:16-24: mean | log_var (+bias), sample = mean + eps*sqrt(exp(lv))
            // C = ml [M,2Z]; p0 = eps [M,Z]; q0 = zs [M,Z]
            if (TN == 2) {
                const int Z = a.gwidth;
                const float mean = v[0] + (a.bias ? pre.f[0] : 0.0f);
                const float lv = v[1 % TN] + (a.bias ? pre.f[1] : 0.0f);
                a.C[(size_t)m * a.ldc + u] = mean;
                a.C[(size_t)m * a.ldc + Z + u] = lv;
                const float zv = mean + pre.f[2] * sqrtf(expf(lv));
                a.q0[(size_t)m * Z + u] = zv;
                if (a.q0_16) a.q0_16[(size_t)m * Z + u] = bf16_of(zv);
            }
        } else if (E == AIR_EPI_LSTM_BWD || E == AIR_EPI_LSTM_BWD_TAIL) {
            // v[0] (+ addend) = d loss / d h'.  p0 = acts, p1 = c_prev, p2 = c, p3 = dc_in (nullable)
            // q0 = dgates [M,4R], q1 = dc_prev [M,R], q2 = dgsum [M,4R] (nullable; i0 = accumulate)
            if (TM == 1) {
                const int R = a.gwidth;
                const bool tail = E == AIR_EPI_LSTM_BWD_TAIL;
                const float addv = a.addend ? pre.f[7] : 0.0f;
                if (tail && m < a.i0) { a.C[(size_t)m * a.ldc + u] = v[0] + addv; continue; }
                const int mrow = tail ? m - a.i0 : m;
                const bool accumulate = tail ? false : (a.i0 != 0);
                const float dhv = v[0] + addv;
                const float si = pre.f[0], tj = pre.f[1], sf = pre.f[2], so = pre.f[3];
                const size_t idx = (size_t)mrow * R + u;
                const float tc = tanhf(pre.f[5]);
                const float dc = (a.p3 ? pre.f[6] : 0.0f) + dhv * so * (1.0f - tc * tc);
                const float dgi = dc * tj * si * (1.0f - si);
                const float dgj = dc * si * (1.0f - tj * tj);
                const float dgf = dc * pre.f[4] * sf * (1.0f - sf);
                const float dgo = dhv * tc * so * (1.0f - so);
                float* dg = a.q0 + (size_t)mrow * 4 * R;
                dg[u] = dgi; dg[R + u] = dgj; dg[2 * R + u] = dgf; dg[3 * R + u] = dgo;
                if (a.q0_16) {
                    unsigned short* dgb = a.q0_16 + (size_t)mrow * 4 * R;
                    dgb[u] = bf16_of(dgi); dgb[R + u] = bf16_of(dgj); dgb[2 * R + u] = bf16_of(dgf); dgb[3 * R + u] = bf16_of(dgo);
                }
                a.q1[idx] = dc * sf;
                if (a.q2) {
                    float* ds = a.q2 + (size_t)mrow * 4 * R;
                    float s0 = dgi, s1 = dgj, s2 = dgf, s3 = dgo;
                    if (accumulate) { s0 = ds[u] + dgi; s1 = ds[R + u] + dgj; s2 = ds[2 * R + u] + dgf; s3 = ds[3 * R + u] + dgo; }
                    ds[u] = s0; ds[R + u] = s1; ds[2 * R + u] = s2; ds[3 * R + u] = s3;
                    if (a.q2_16) {       // the running sum's twin: passed by the caller with the LAST accumulation only
                        unsigned short* dsb = a.q2_16 + (size_t)mrow * 4 * R;
                        dsb[u] = bf16_of(s0); dsb[R + u] = bf16_of(s1); dsb[2 * R + u] = bf16_of(s2); dsb[3 * R + u] = bf16_of(s3);
                    }
                }
            }
        } else if (E == AIR_EPI_REPARAM_BWD) {
            // v[0] = d loss / d z-sample.  p0 = ml [M,2Z], p1 = eps, p2 = att (mask), p3 = dyn; C = d_ml [M,2Z]
            const int Z = a.gwidth;
            const float klg = pre.f[3] * pre.f[4];
            const float pv = pre.f[5], pm = pre.f[6];
            const float var = expf(pre.f[1]);
            const float sd = sqrtf(var);
            const float d = v[0];
            const float dmean = d + klg * (pre.f[0] - pm) / pv;
            const float dlv = d * pre.f[2] * 0.5f * sd + klg * 0.5f * (var / pv - 1.0f);
            a.C[(size_t)m * a.ldc + u] = dmean;
            a.C[(size_t)m * a.ldc + Z + u] = dlv;
            if (a.C16) { a.C16[(size_t)m * a.ldc + u] = bf16_of(dmean); a.C16[(size_t)m * a.ldc + Z + u] = bf16_of(dlv); }
        }
    }
}

// cross-wave (split-K) reduction in a fixed order: ((w0 + w1) + w2) + w3, result in Red region 0
template <int TM, int TN>
__device__ __forceinline__ void reduce_waves(f32x4 (&acc)[TM][TN], float* Red, int lane, int wave) {
    constexpr int RS = TM * TN * 4 * 64;
    if (wave > 0) {
        float* r = Red + (wave - 1) * RS;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) r[((i * TN + j) * 4 + q) * 64 + lane] = acc[i][j][q];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = ((i * TN + j) * 4 + q) * 64 + lane;
                    float v = acc[i][j][q];
#pragma unroll
                    for (int w = 0; w < 3; ++w) v += Red[w * RS + o];
                    Red[o] = v;
                }
    }
    __syncthreads();
}

// XCD-aware workgroup -> tile map.  Workgroup b is observed to run on XCD b % 8, each XCD with a
// private L2: the default map would spread the m-tiles that share one panel of weight columns
// over all eight L2s (measured: 3-5x the algorithmic HBM-side traffic).  Here every XCD takes a
// contiguous chunk of the tile list, ordered m-fastest, so a weight panel is fetched into ONE L2
// and its other users hit there (bijective for any grid size; a speed choice only, never relied
// on for correctness).
__device__ __forceinline__ void xcd_tile(int& tile_m, int& tile_n) {
    const int nx = gridDim.x, ny = gridDim.y, nwg = nx * ny;
    const int bid = blockIdx.y * nx + blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tile_m = swz % ny;
    tile_n = swz / ny;
}

__host__ __device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace airg
