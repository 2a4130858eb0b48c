// Small-batch GEMM with fused epilogues for the AIR loop (LSTM / heads / VAE
// MatMul + BiasAdd + activation, their data- and weight-gradients).
//
// Shapes on this path are skinny (M = batch = 64..256, or the contraction is
// N_steps*batch = 192), so the kernel is built for LATENCY, not peak MFMA rate:
//   * one workgroup = 4 waves = one (16*TM x 16*TN) output tile.  The whole
//     K-panel of both operands (<= ~1024 deep) is staged in LDS with ONE
//     barrier: every thread issues all of its 16-byte global loads up front
//     (one memory latency), the four waves then run their MFMA chains
//     back-to-back on interleaved k-steps (one wave per SIMD = four matrix
//     pipes on the same tile) and are summed through LDS in a fixed order
//     (deterministic, no atomics);
//   * lane->element maps of the staging stores are chosen per layout so that
//     the transposing LDS stores and the fragment reads are bank-conflict free
//     (k-major images, row strides = 17 or = 16 mod 32 dwords);
//   * "grouped" column tiles (tile j covers columns n0 + j*gstride ..+16) put
//     the four LSTM gates / the (mean, log-variance) pair of one unit in the
//     same lane, so the pointwise LSTM / re-parameterisation math is fused into
//     the GEMM epilogue (no extra launches, no round trip through HBM);
//   * grid.z splits K into slabs for the one deep contraction (x.Wx, K = 2500);
//     the consumer (LSTM epilogue) sums the slabs in fixed order;
//   * precision 0: v_mfma_f32_16x16x4_f32 (exact fp32, bit-equal to an fmaf
//     chain); precision 1: operands rounded to bf16 while staging,
//     v_mfma_f32_16x16x32_bf16, fp32 accumulate.
#include "air_gemm_common.h"
#include <cstdlib>
#include <type_traits>
#include <cstdio>

AIR_STAMPS_READER(air_debug_stamps_gemm)

using namespace airg;

namespace airg {
// air_gemm_bf16.hip: the bf16-twin-operand kernels
int twin_rounds(const Args& a, int tm, int tn, bool ta, bool tb);
int twin_launch(const Args& a, int tm, int tn, bool tb, dim3 grid, hipStream_t s);
void twin_kernel_name(const Args& a, int tm, int tn, bool tb, char* buf, int n);
int xw_tp_ok(const Args& a, int precision, bool ta, bool tb, int ksplit);
int xw_tp_launch(const Args& a, int job_planes_hint, hipStream_t s);
int xw_tp_columns(const Args& a);
}

namespace {

// ---------------------------------------------------------------------------
// fp32: k-major LDS images As[k][LA], Bs[k][LB], double-buffered chunks of BKC.
// A frag: lane l holds A[m = l&15][k = l>>4]; B frag: B[k = l>>4][n = l&15].
// ALL 16-byte global loads of up to NCH chunks are issued before the first one
// is consumed: a workgroup pays ONE memory round trip (~2 us when the producer
// kernel ran on another XCD), then streams chunk by chunk through LDS with one
// barrier each while the later loads are still landing.
// ---------------------------------------------------------------------------
template <int TM, int TN, bool TA, bool TB>
__global__ __launch_bounds__(THREADS) void gemm_f32_kernel(Args a)
{
    constexpr int EPI_ = -1;                               // fallback kernel: epilogue chosen at run time
    constexpr int BM = 16 * TM, BN = 16 * TN;
    constexpr int BKC = (TM * TN >= 4) ? 64 : 128;
    // transposing stores (NN-A, NT-B) want an odd stride; direct 16-byte stores want stride = 16 (mod 32)
    constexpr int LA = TA ? (BM == 16 ? 16 : BM + 16) : BM + 1;
    constexpr int LB = TB ? BN + 1 : (BN == 16 ? 16 : BN + 16);
    constexpr int PA = BM * BKC / 1024, PB = BN * BKC / 1024;     // float4 per thread per chunk
    constexpr int NCHR = 32 / (PA + PB);
    constexpr int NCH = NCHR < 1 ? 1 : (NCHR > 8 ? 8 : NCHR);     // chunks in flight (<= 128 VGPRs of data)
    __shared__ __attribute__((aligned(16))) float As[2][BKC * LA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BKC * LB];
    __shared__ float Red[3 * TM * TN * 4 * 64];

    if ((int)blockIdx.z < a.job_on) {                    // block-uniform: the prologue's planes of workgroups (dispatched first)
        const long plane = (long)gridDim.x * gridDim.y;
        air_step_job_run(a.job, blockIdx.z * plane + (long)blockIdx.y * gridDim.x + blockIdx.x, plane * a.job_on);
        return;
    }
    const int nslab = (int)gridDim.z - a.job_on;
    const int zslab = (int)blockIdx.z - a.job_on;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tile_m, tile_n;
    xcd_tile(tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN / TN * (a.gstride == 16 ? TN : 1);
    const int kbeg = zslab * a.kslab;
    const int kend = min(a.K, kbeg + a.kslab);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool vecA = ((a.lda & 3) == 0) && aligned16(a.A);
    const bool vecB = ((a.ldb & 3) == 0) && aligned16(a.B) && ((a.gstride & 3) == 0);
    Pre<TM, TN> pre;
    if (nslab == 1) epilogue_prefetch<TM, TN, EPI_>(a, pre, m0, n0, lane, wave);

    auto loadA = [&](int k0, int i) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!TA) {
            // A[m*lda + k]: lane map (m_lo = l&3, k4_lo = l>>2) -> conflict-free transposing stores
            const int g = wave + 4 * i;
            const int m = (g % (BM / 4)) * 4 + (lane & 3);
            const int k4 = (g / (BM / 4)) * 16 + (lane >> 2);
            const int gm = m0 + m, gk = k0 + k4 * 4;
            if (gm < a.M && gk < kend) {
                const float* src = a.A + (size_t)gm * a.lda + gk;
                if (vecA && gk + 3 < kend) v = *reinterpret_cast<const float4*>(src);
                else {
                    v.x = src[0];
                    if (gk + 1 < kend) v.y = src[1];
                    if (gk + 2 < kend) v.z = src[2];
                    if (gk + 3 < kend) v.w = src[3];
                }
            }
        } else {
            // A[k*lda + m]: 16-byte loads along m
            constexpr int MQ = BM / 4;
            const int it = tid + THREADS * i;
            const int k = it / MQ, mq = it % MQ;
            const int gm = m0 + mq * 4, gk = k0 + k;
            if (gk < kend && gm < a.M) {
                const float* src = a.A + (size_t)gk * a.lda + gm;
                if (vecA && gm + 3 < a.M) v = *reinterpret_cast<const float4*>(src);
                else {
                    v.x = src[0];
                    if (gm + 1 < a.M) v.y = src[1];
                    if (gm + 2 < a.M) v.z = src[2];
                    if (gm + 3 < a.M) v.w = src[3];
                }
            }
        }
        return v;
    };
    auto storeA = [&](float* as, int i, const float4& v) {
        if (!TA) {
            const int g = wave + 4 * i;
            const int m = (g % (BM / 4)) * 4 + (lane & 3);
            const int k4 = (g / (BM / 4)) * 16 + (lane >> 2);
            float* d = as + (size_t)(k4 * 4) * LA + m;
            d[0] = v.x; d[LA] = v.y; d[2 * LA] = v.z; d[3 * LA] = v.w;
        } else {
            constexpr int MQ = BM / 4;
            const int it = tid + THREADS * i;
            *reinterpret_cast<float4*>(as + (size_t)(it / MQ) * LA + (it % MQ) * 4) = v;
        }
    };
    auto loadB = [&](int k0, int i) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (TB) {
            // B[n*ldb + k] (k contiguous): same map as NN-A
            const int g = wave + 4 * i;
            const int n = (g % (BN / 4)) * 4 + (lane & 3);
            const int k4 = (g / (BN / 4)) * 16 + (lane >> 2);
            const int gn = n0 + n, gk = k0 + k4 * 4;
            if (gn < a.N && gk < kend) {
                const float* src = a.B + (size_t)gn * a.ldb + gk;
                if (vecB && gk + 3 < kend) v = *reinterpret_cast<const float4*>(src);
                else {
                    v.x = src[0];
                    if (gk + 1 < kend) v.y = src[1];
                    if (gk + 2 < kend) v.z = src[2];
                    if (gk + 3 < kend) v.w = src[3];
                }
            }
        } else {
            // B[k*ldb + n]: 16-byte loads along n (per 16-column group)
            constexpr int NQ = BN / 4;
            const int it = tid + THREADS * i;
            const int k = it / NQ, q4 = it % NQ;
            const int j = q4 >> 2, c = (q4 & 3) * 4;
            const int cg = n0 + c + (a.gstride == 16 ? j * 16 : 0);   // bound-check coordinate
            const int gn = n0 + j * a.gstride + c, gk = k0 + k;
            if (gk < kend) {
                const float* src = a.B + (size_t)gk * a.ldb + gn;
                const int lim = min(a.gwidth - cg, a.N - gn);        // valid columns from here
                if (vecB && lim >= 4) v = *reinterpret_cast<const float4*>(src);
                else {
                    if (lim > 0) v.x = src[0];
                    if (lim > 1) v.y = src[1];
                    if (lim > 2) v.z = src[2];
                    if (lim > 3) v.w = src[3];
                }
            }
        }
        return v;
    };
    auto storeB = [&](float* bs, int i, const float4& v) {
        if (TB) {
            const int g = wave + 4 * i;
            const int n = (g % (BN / 4)) * 4 + (lane & 3);
            const int k4 = (g / (BN / 4)) * 16 + (lane >> 2);
            float* d = bs + (size_t)(k4 * 4) * LB + n;
            d[0] = v.x; d[LB] = v.y; d[2 * LB] = v.z; d[3 * LB] = v.w;
        } else {
            constexpr int NQ = BN / 4;
            const int it = tid + THREADS * i;
            *reinterpret_cast<float4*>(bs + (size_t)(it / NQ) * LB + (it % NQ) * 4) = v;
        }
    };

    for (int ks0 = kbeg; ks0 < kend; ks0 += NCH * BKC) {
        if (ks0 > kbeg) __syncthreads();
        float4 va[NCH][PA], vb[NCH][PB];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k0 = ks0 + c * BKC;
            if (k0 < kend) {                                         // block-uniform
#pragma unroll
                for (int i = 0; i < PA; ++i) va[c][i] = loadA(k0, i);
#pragma unroll
                for (int i = 0; i < PB; ++i) vb[c][i] = loadB(k0, i);
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k0 = ks0 + c * BKC;
            if (k0 < kend) {
                float* as = As[c & 1];
                float* bs = Bs[c & 1];
#pragma unroll
                for (int i = 0; i < PA; ++i) storeA(as, i, va[c][i]);
#pragma unroll
                for (int i = 0; i < PB; ++i) storeB(bs, i, vb[c][i]);
                __syncthreads();         // chunk visible; every wave is past the MFMAs of chunk c-1
                const int steps = (min(BKC, kend - k0) + 3) >> 2;
                constexpr int SU = BKC / 16;                         // k-steps per wave per chunk
                float av[SU][TM], bv[SU][TN];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int sidx = wave + 4 * u;
                    const bool ok = sidx < steps;                    // wave-uniform
                    const int kk = (ok ? sidx : 0) * 4 + (lane >> 4);
#pragma unroll
                    for (int i = 0; i < TM; ++i) { const float t = as[(size_t)kk * LA + i * 16 + (lane & 15)]; av[u][i] = ok ? t : 0.0f; }
#pragma unroll
                    for (int j = 0; j < TN; ++j) { const float t = bs[(size_t)kk * LB + j * 16 + (lane & 15)]; bv[u][j] = ok ? t : 0.0f; }
                }
#pragma unroll
                for (int u = 0; u < SU; ++u)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
            }
        }
    }
    reduce_waves<TM, TN>(acc, Red, lane, wave);
    if (nslab > 1) {
        // split-K slab: plain store, the consumer sums the slabs
        float* Cz = a.C + (size_t)zslab * a.slab_stride;
        for (int t = wave; t < TM * TN; t += 4) {
            const int i = t / TN, j = t % TN;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + i * 16 + (lane >> 4) * 4 + q;
                const int n = n0 + j * 16 + (lane & 15);
                if (m < a.M && n < a.N) Cz[(size_t)m * a.ldc + n] = Red[(t * 4 + q) * 64 + lane];
            }
        }
        return;
    }
    epilogue<TM, TN, EPI_>(a, pre, Red, m0, n0, lane, wave);
}

// ---------------------------------------------------------------------------
// bf16 path: 128-deep K chunks, operands rounded to bf16 while staging into
// [m][k] / [n][k] images (k contiguous, one 16-byte fragment read per MFMA).
// ---------------------------------------------------------------------------
template <int TM, int TN, bool TA, bool TB>
__global__ __launch_bounds__(THREADS) void gemm_bf16_kernel(Args a)
{
    constexpr int BK = 128;
    constexpr int BM = 16 * TM, BN = 16 * TN;
    constexpr int LK = BK + 8;                       // halves per row
    constexpr int NA = BM * BK / THREADS;
    constexpr int NB = BN * BK / THREADS;
    __shared__ __attribute__((aligned(16))) unsigned short As[BM * LK];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[BN * LK];
    __shared__ float Red[3 * TM * TN * 4 * 64];

    if ((int)blockIdx.z < a.job_on) {                    // block-uniform: the prologue's planes of workgroups (dispatched first)
        const long plane = (long)gridDim.x * gridDim.y;
        air_step_job_run(a.job, blockIdx.z * plane + (long)blockIdx.y * gridDim.x + blockIdx.x, plane * a.job_on);
        return;
    }
    const int nslab = (int)gridDim.z - a.job_on;
    const int zslab = (int)blockIdx.z - a.job_on;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tile_m, tile_n;
    xcd_tile(tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN / TN * (a.gstride == 16 ? TN : 1);
    const int kbeg = zslab * a.kslab;
    const int kend = min(a.K, kbeg + a.kslab);

    float ra[NA], rb[NB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * THREADS;
            int m, k;
            if (TA) { k = idx / BM; m = idx % BM; } else { m = idx / BK; k = idx % BK; }
            const int gm = m0 + m, gk = k0 + k;
            float v = 0.0f;
            if (gm < a.M && gk < kend) v = TA ? a.A[(size_t)gk * a.lda + gm] : a.A[(size_t)gm * a.lda + gk];
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + i * THREADS;
            int n, k;
            if (TB) { n = idx / BK; k = idx % BK; } else { k = idx / BN; n = idx % BN; }
            const int j = n >> 4, c = n & 15;
            const int cg = n0 + c + (a.gstride == 16 ? j * 16 : 0);
            const int gn = n0 + j * a.gstride + c, gk = k0 + k;
            float v = 0.0f;
            if (cg < a.gwidth && gn < a.N && gk < kend)
                v = TB ? a.B[(size_t)gn * a.ldb + gk] : a.B[(size_t)gk * a.ldb + gn];
            rb[i] = v;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * THREADS;
            int m, k;
            if (TA) { k = idx / BM; m = idx % BM; } else { m = idx / BK; k = idx % BK; }
            As[m * LK + k] = f32_to_bf16_rne(ra[i]);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + i * THREADS;
            int n, k;
            if (TB) { n = idx / BK; k = idx % BK; } else { k = idx / BN; n = idx % BN; }
            Bs[n * LK + k] = f32_to_bf16_rne(rb[i]);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
        stage();
        __syncthreads();
        if (k0 + BK < kend) fetch(k0 + BK);
        const int kk = wave * 32 + (lane >> 4) * 8;   // wave w owns k in [w*32, w*32+32): one 16x16x32 step
        bf16x8 av[TM], bv[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const bf16x8*>(&As[(i * 16 + (lane & 15)) * LK + kk]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(&Bs[(j * 16 + (lane & 15)) * LK + kk]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    reduce_waves<TM, TN>(acc, Red, lane, wave);
    if (nslab > 1) {
        float* Cz = a.C + (size_t)zslab * a.slab_stride;
        for (int t = wave; t < TM * TN; t += 4) {
            const int i = t / TN, j = t % TN;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + i * 16 + (lane >> 4) * 4 + q;
                const int n = n0 + j * 16 + (lane & 15);
                if (m < a.M && n < a.N) Cz[(size_t)m * a.ldc + n] = Red[(t * 4 + q) * 64 + lane];
            }
        }
        return;
    }
    Pre<TM, TN> pre;
    epilogue_prefetch<TM, TN, -1>(a, pre, m0, n0, lane, wave);
    epilogue<TM, TN, -1>(a, pre, Red, m0, n0, lane, wave);
}


// ---------------------------------------------------------------------------
// bf16 path, lean variant (A row-major, 16-byte aligned operands, K % 4 == 0;
// NN additionally N % 4 == 0).  K is cut into images of 64; every operand image
// lives in LDS as [row or column][64 k] bf16 whose eight 16-byte slots are
// XOR-swizzled by the row index: conflict-free for the 8-lane ds_write_b128
// groups and the 16-lane ds_read_b128 fragment groups.  Rounding to bf16 (RNE,
// v_cvt_pk_bf16_f32) happens on the way into LDS.  All global loads of a round
// of up to R images are issued back to back (uniform base + 32-bit offsets, no
// divergent bounds branches: out-of-range elements are redirected to offset 0
// and zeroed by a select), then ONE barrier, then the four waves take the
// images of the round cyclically -- one ds_read_b128 per MFMA operand.
// ---------------------------------------------------------------------------
// (tried: raw buffer loads with the hardware range check standing in for the EXEC-masked branches --
// same results, no gain on the 16x16-tile kernels and 8.5 -> 11.7 us on the K = 2500 one)
// (the zeroing is a bit mask, not a select: where the loaded registers are shuffled before their first use -- the NN-B
// transposition of the fp32 kernel -- the compiler turned `ok ? t : 0` into an EXEC-masked branch around the load AND the
// shuffle, with an s_waitcnt vmcnt(0) inside: eight loads of a task completed one after the other, ~4 us of an 8.5 us launch)
__device__ __forceinline__ float4 ldg16(const char* base, unsigned off, bool ok) {
    const uint4 t = *reinterpret_cast<const uint4*>(base + (ok ? off : 0u));
    const unsigned m = ok ? 0xffffffffu : 0u;
    return make_float4(__uint_as_float(t.x & m), __uint_as_float(t.y & m), __uint_as_float(t.z & m), __uint_as_float(t.w & m));
}

__device__ __forceinline__ float2 ldg8(const char* base, unsigned off, bool ok) {
    const uint2 t = *reinterpret_cast<const uint2*>(base + (ok ? off : 0u));
    const unsigned m = ok ? 0xffffffffu : 0u;
    return make_float2(__uint_as_float(t.x & m), __uint_as_float(t.y & m));
}
// 16 bytes as one load, or as two 8-byte loads when the operand is only 8-byte aligned
// (row strides / column-group strides that are even but not multiples of 4: Z = 50)
// HALF is a COMPILE-TIME choice: a run-time one puts every load in its own if/else whose join needs the
// loaded value, i.e. an s_waitcnt vmcnt(0) behind each load -- the loads of a round then complete one
// after the other instead of all being in flight (measured: 1.3 us of a 3.0 us kernel body).
template <bool HALF>
__device__ __forceinline__ float4 ldg16x(const char* base, unsigned off, bool ok_lo, bool ok_hi) {
    if (!HALF) return ldg16(base, off, ok_lo);
    const float2 lo = ldg8(base, off, ok_lo), hi = ldg8(base, off + 8u, ok_hi);
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}

template <int TM, int TN>
struct V2Cfg {
    static constexpr int BM = 16 * TM, BN = 16 * TN, KB = 64;
    static constexpr int IMG = (BM + BN) * KB;                           // bf16 elements per image pair
    static constexpr int RL = 49152 / (IMG * 2);                         // LDS: <= 48 KB of images
    static constexpr int RV = 28 / (TM + TN);                            // VGPRs: <= ~28 float4 in flight
    static constexpr int R0 = RL < RV ? RL : RV;
    static constexpr int R = R0 > 8 ? 8 : (R0 < 1 ? 1 : R0);
    static constexpr int RED = 3 * TM * TN * 4 * 64 * 4;                 // bytes of the cross-wave reduction
    static constexpr int BYTES = (R * IMG * 2 > RED) ? R * IMG * 2 : RED;
};

template <int TM, int TN, bool TB, int EPI_>
__global__ __launch_bounds__(THREADS) void gemm_bf16v2_kernel(Args a)
{
    using Cfg = V2Cfg<TM, TN>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, KB = Cfg::KB, R = Cfg::R;
    __shared__ __attribute__((aligned(16))) unsigned char Lds[Cfg::BYTES];
    unsigned short* ImgA = reinterpret_cast<unsigned short*>(Lds);       // [R][BM][64]
    unsigned short* ImgB = ImgA + R * BM * KB;                           // [R][BN][64]
    float* Red = reinterpret_cast<float*>(Lds);

    if ((int)blockIdx.z < a.job_on) {                    // block-uniform: the prologue's planes of workgroups (dispatched first)
        const long plane = (long)gridDim.x * gridDim.y;
        air_step_job_run(a.job, blockIdx.z * plane + (long)blockIdx.y * gridDim.x + blockIdx.x, plane * a.job_on);
        return;
    }
    const int nslab = (int)gridDim.z - a.job_on;
    const int zslab = (int)blockIdx.z - a.job_on;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tile_m, tile_n;
    xcd_tile(tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = EPI_ == AIR_EPI_LSTM_FWD0 ? tile_n * 4 : tile_n * BN / TN * (a.gstride == 16 ? TN : 1);
    const int kbeg = zslab * a.kslab;
    const int kend = min(a.K, kbeg + a.kslab);

    AIR_STAMP(56);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Pre<TM, TN> pre;

    const char* Ab = reinterpret_cast<const char*>(a.A);
    const char* Bb = reinterpret_cast<const char*>(a.B);
    const bool a8 = !(((a.lda & 3) == 0) && aligned16(a.A) && ((a.K & 3) == 0));
    const bool b8 = !(((a.ldb & 3) == 0) && aligned16(a.B) &&
                      (TB ? ((a.K & 3) == 0) : (((a.N & 3) == 0) && ((a.gstride & 3) == 0) && ((a.gwidth & 3) == 0))));
    // tasks of one round.  k-contiguous operand (A, and B when TB): task = (image, row, 16-byte chunk of the row's
    // 64 k) -- consecutive lanes read consecutive 16 bytes, so every load instruction covers whole cache lines
    // (a lane that read one k-run of 8 as two float4 left every second 16 bytes of its lines to the other load);
    // NN B: task = (image, column quad, k-run g) = the same 4 columns of 8 consecutive rows = 8 float4.
    constexpr int TA_N = (R * BM * 8 + THREADS - 1) / THREADS;           // A tasks per thread
    constexpr int TBK_N = (R * BN * 8 + THREADS - 1) / THREADS;          // NT-B tasks per thread
    constexpr int TBN_N = (R * BN * 2 + THREADS - 1) / THREADS;          // NN-B tasks per thread

    float4 va[TA_N][2];
    float4 vbk[TB ? TBK_N : 1][2];
    float4 vbn[TB ? 1 : TBN_N][8];
    // ---- every load of one round, issued back to back
    auto issue_loads_t = [&](int kr, auto ha_t, auto hb_t) __attribute__((always_inline)) {
        constexpr bool HA = decltype(ha_t)::value, HB = decltype(hb_t)::value;
#pragma unroll
        for (int i = 0; i < 2 * TA_N; ++i) {
            const int u = tid + THREADS * i;
            const int c = u / (BM * 16), row = (u / 16) % BM, hh = u & 15;
            const int gm = m0 + row, gk = kr + c * KB + hh * 4;
            const bool okr = (u < R * BM * 16) && gm < a.M;
            const unsigned off = ((unsigned)gm * (unsigned)a.lda + (unsigned)gk) * 4u;
            va[i >> 1][i & 1] = ldg16x<HA>(Ab, off, okr && gk < kend, okr && gk + 2 < kend);
        }
        if (TB) {
#pragma unroll
            for (int i = 0; i < 2 * TBK_N; ++i) {
                const int u = tid + THREADS * i;
                const int c = u / (BN * 16), col = (u / 16) % BN, hh = u & 15;
                const int j = col >> 4, cc = col & 15;
                const int gn = n0 + j * a.gstride + cc, cg = n0 + cc + (a.gstride == 16 ? j * 16 : 0);
                const int gk = kr + c * KB + hh * 4;
                const bool okr = (u < R * BN * 16) && cg < a.gwidth && gn < a.N;
                const unsigned off = ((unsigned)gn * (unsigned)a.ldb + (unsigned)gk) * 4u;
                vbk[i >> 1][i & 1] = ldg16x<HB>(Bb, off, okr && gk < kend, okr && gk + 2 < kend);
            }
        } else {
#pragma unroll
            for (int i = 0; i < TBN_N; ++i) {
                const int t = tid + THREADS * i;
                const int c = t / (BN * 2), q = (t / 8) % (BN / 4), g = t & 7;
                const int col = q * 4, j = col >> 4, cc = col & 15;
                // (LSTM_FWD0: column quad q = gate q of the four units n0 .. n0+3)
                const int gn = EPI_ == AIR_EPI_LSTM_FWD0 ? n0 + q * a.gstride : n0 + j * a.gstride + cc;
                const int cg = EPI_ == AIR_EPI_LSTM_FWD0 ? n0 : n0 + cc + (a.gstride == 16 ? j * 16 : 0);
                const int gk = kr + c * KB + g * 8;
                const bool okc = (t < R * BN * 2) && cg < a.gwidth && gn < a.N;
                const bool okh = okc && cg + 2 < a.gwidth && gn + 2 < a.N;     // upper half of the column quad
                const unsigned off = ((unsigned)gk * (unsigned)a.ldb + (unsigned)gn) * 4u;
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    vbn[i][r] = ldg16x<HB>(Bb, off + (unsigned)r * ((unsigned)a.ldb * 4u), okc && gk + r < kend,
                                           okh && gk + r < kend);
            }
        }
    };
    // software pipeline over rounds: the loads of round r+1 are issued right after the registers of
    // round r were drained into LDS, so their latency runs under the barrier and the MFMAs of round r.
    // The whole K loop is instantiated per operand alignment and picked by ONE uniform branch, so that
    // no control-flow join sits between a round's loads and their first use.
    auto k_loop = [&](auto ha_t, auto hb_t) __attribute__((always_inline)) {
    auto issue_loads = [&](int kr) __attribute__((always_inline)) { issue_loads_t(kr, ha_t, hb_t); };
    issue_loads(kbeg);
    // the epilogue's operands ride behind the first round's panels (same memory round trip)
    if (nslab == 1) epilogue_prefetch<TM, TN, EPI_>(a, pre, m0, n0, lane, wave);
    AIR_STAMP(57);
    for (int kr = kbeg; kr < kend; kr += R * KB) {
        if (kr > kbeg) __syncthreads();                                   // images of the previous round consumed
        // ---- round to bf16 and store the images
#pragma unroll
        for (int i = 0; i < 2 * TA_N; ++i) {
            const int u = tid + THREADS * i;
            const int c = u / (BM * 16), row = (u / 16) % BM, hh = u & 15;
            const float4& x = va[i >> 1][i & 1];
            uint2 w;
            w.x = pack_bf16(x.x, x.y); w.y = pack_bf16(x.z, x.w);
            // half of the swizzled 16-byte slot of k-run hh >> 1
            if (u < R * BM * 16)
                *reinterpret_cast<uint2*>(&ImgA[(c * BM + row) * KB + (((hh >> 1) ^ (row & 7)) << 3) + (hh & 1) * 4]) = w;
        }
        if (TB) {
#pragma unroll
            for (int i = 0; i < 2 * TBK_N; ++i) {
                const int u = tid + THREADS * i;
                const int c = u / (BN * 16), col = (u / 16) % BN, hh = u & 15;
                const float4& x = vbk[i >> 1][i & 1];
                uint2 w;
                w.x = pack_bf16(x.x, x.y); w.y = pack_bf16(x.z, x.w);
                if (u < R * BN * 16)
                    *reinterpret_cast<uint2*>(&ImgB[(c * BN + col) * KB + (((hh >> 1) ^ (col & 7)) << 3) + (hh & 1) * 4]) = w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < TBN_N; ++i) {
                const int t = tid + THREADS * i;
                const int c = t / (BN * 2), q = (t / 8) % (BN / 4), g = t & 7;
                if (t < R * BN * 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int col = q * 4 + e;
                        auto el = [&](int r) { const float4& x = vbn[i][r]; return e == 0 ? x.x : e == 1 ? x.y : e == 2 ? x.z : x.w; };
                        uint4 w;
                        w.x = pack_bf16(el(0), el(1)); w.y = pack_bf16(el(2), el(3));
                        w.z = pack_bf16(el(4), el(5)); w.w = pack_bf16(el(6), el(7));
                        *reinterpret_cast<uint4*>(&ImgB[(c * BN + col) * KB + ((g ^ (col & 7)) << 3)]) = w;
                    }
                }
            }
        }
        if (kr + R * KB < kend) issue_loads(kr + R * KB);
        __syncthreads();
        AIR_STAMP(58);
        // ---- MFMAs: wave w owns the images whose index WITHIN THE SLAB is w (mod 4) -- independent of the
        // round length R, so every tile configuration (and the twin-operand kernels) sums k in the same order
        const int cfirst = (wave - ((kr - kbeg) / KB)) & 3;
#pragma unroll
        for (int cc = 0; cc < (R + 3) / 4; ++cc) {
            const int c = cfirst + 4 * cc;
            if (c < R && kr + c * KB < kend) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int slot = ks * 4 + (lane >> 4);
                    bf16x8 av[TM], bv[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = i * 16 + (lane & 15);
                        av[i] = *reinterpret_cast<const bf16x8*>(&ImgA[(c * BM + row) * KB + ((slot ^ (row & 7)) << 3)]);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = j * 16 + (lane & 15);
                        bv[j] = *reinterpret_cast<const bf16x8*>(&ImgB[(c * BN + col) * KB + ((slot ^ (col & 7)) << 3)]);
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    };
    {
        using T_ = std::true_type; using F_ = std::false_type;
        if (!a8 && !b8) k_loop(F_{}, F_{});
        else if (!a8) k_loop(F_{}, T_{});
        else if (!b8) k_loop(T_{}, F_{});
        else k_loop(T_{}, T_{});
    }
    AIR_STAMP(59);
    __syncthreads();                                                      // Red aliases the images
    reduce_waves<TM, TN>(acc, Red, lane, wave);
    AIR_STAMP(60);
    if (nslab > 1) {
        float* Cz = a.C + (size_t)zslab * a.slab_stride;
        for (int t = wave; t < TM * TN; t += 4) {
            const int i = t / TN, j = t % TN;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + i * 16 + (lane >> 4) * 4 + q;
                const int n = n0 + j * 16 + (lane & 15);
                if (m < a.M && n < a.N) Cz[(size_t)m * a.ldc + n] = Red[(t * 4 + q) * 64 + lane];
            }
        }
        return;
    }
    epilogue<TM, TN, EPI_>(a, pre, Red, m0, n0, lane, wave);
    AIR_STAMP(61);
}



// fp32 twin of the lean kernel: the same images with fp32 elements (16 swizzled 16-byte slots per
// row), exact fp32 products on v_mfma_f32_16x16x4_f32; dynamic LDS (up to 80 KB).
template <int TM, int TN>
struct F32V2Cfg {
    static constexpr int BM = 16 * TM, BN = 16 * TN, KB = 64;
    static constexpr int IMG = (BM + BN) * KB * 4;                       // bytes per image pair
    static constexpr int RL = 81920 / IMG;
    static constexpr int RV = 24 / (TM + TN);
    static constexpr int R0 = RL < RV ? RL : RV;
    static constexpr int R = R0 > 8 ? 8 : (R0 < 1 ? 1 : R0);
    static constexpr int RED = 3 * TM * TN * 4 * 64 * 4;
    static constexpr int BYTES = (R * IMG > RED) ? R * IMG : RED;
};

template <int TM, int TN, bool TB, int EPI_>
__global__ __launch_bounds__(THREADS) void gemm_f32v2_kernel(Args a)
{
    using Cfg = F32V2Cfg<TM, TN>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, KB = Cfg::KB, R = Cfg::R;
    extern __shared__ __attribute__((aligned(16))) unsigned char Lds[];  // Cfg::BYTES (may exceed 64 KB)
    float* ImgA = reinterpret_cast<float*>(Lds);                         // [R][BM][64]
    float* ImgB = ImgA + R * BM * KB;                                    // [R][BN][64]
    float* Red = reinterpret_cast<float*>(Lds);

    if ((int)blockIdx.z < a.job_on) {                    // block-uniform: the prologue's planes of workgroups (dispatched first)
        const long plane = (long)gridDim.x * gridDim.y;
        air_step_job_run(a.job, blockIdx.z * plane + (long)blockIdx.y * gridDim.x + blockIdx.x, plane * a.job_on);
        return;
    }
    const int nslab = (int)gridDim.z - a.job_on;
    const int zslab = (int)blockIdx.z - a.job_on;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tile_m, tile_n;
    xcd_tile(tile_m, tile_n);
    // 16 columns = 4 gates x 4 units: the first-step launch, and (EPI_LSTM_FWD_Q) the later LSTM steps -- 256 workgroups
    // of 16 columns instead of 64 of 64, the same accumulation per element
    constexpr bool QUADF = EPI_ == AIR_EPI_LSTM_FWD0 || EPI_ == EPI_LSTM_FWD_Q;
    const int m0 = tile_m * BM, n0 = QUADF ? tile_n * 4 : tile_n * BN / TN * (a.gstride == 16 ? TN : 1);
    const int kbeg = zslab * a.kslab;
    const int kend = min(a.K, kbeg + a.kslab);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Pre<TM, TN> pre;
    if (nslab == 1) epilogue_prefetch<TM, TN, EPI_>(a, pre, m0, n0, lane, wave);

    const char* Ab = reinterpret_cast<const char*>(a.A);
    const char* Bb = reinterpret_cast<const char*>(a.B);
    const bool a8 = !(((a.lda & 3) == 0) && aligned16(a.A) && ((a.K & 3) == 0));
    const bool b8 = !(((a.ldb & 3) == 0) && aligned16(a.B) &&
                      (TB ? ((a.K & 3) == 0) : (((a.N & 3) == 0) && ((a.gstride & 3) == 0) && ((a.gwidth & 3) == 0))));
    // tasks of one round.  k-contiguous operand (A, and B when TB): task = (image, row, k-run g) = 2 float4;
    // NN B: task = (image, column quad, k-run g) = the same 4 columns of 8 consecutive rows = 8 float4.
    constexpr int TA_N = (R * BM * 8 + THREADS - 1) / THREADS;           // A tasks per thread
    constexpr int TBK_N = (R * BN * 8 + THREADS - 1) / THREADS;          // NT-B tasks per thread
    constexpr int TBN_N = (R * BN * 2 + THREADS - 1) / THREADS;          // NN-B tasks per thread

    float4 va[TA_N][2];
    float4 vbk[TB ? TBK_N : 1][2];
    float4 vbn[TB ? 1 : TBN_N][8];
    // ---- every load of one round, issued back to back
    auto issue_loads_t = [&](int kr, auto ha_t, auto hb_t) __attribute__((always_inline)) {
        constexpr bool HA = decltype(ha_t)::value, HB = decltype(hb_t)::value;
#pragma unroll
        for (int i = 0; i < TA_N; ++i) {
            const int t = tid + THREADS * i;
            const int c = t / (BM * 8), row = (t / 8) % BM, g = t & 7;
            const int gm = m0 + row, gk = kr + c * KB + g * 8;
            const bool okr = (t < R * BM * 8) && gm < a.M;
            const unsigned off = ((unsigned)gm * (unsigned)a.lda + (unsigned)gk) * 4u;
            va[i][0] = ldg16x<HA>(Ab, off, okr && gk < kend, okr && gk + 2 < kend);
            va[i][1] = ldg16x<HA>(Ab, off + 16u, okr && gk + 4 < kend, okr && gk + 6 < kend);
        }
        if (TB) {
#pragma unroll
            for (int i = 0; i < TBK_N; ++i) {
                const int t = tid + THREADS * i;
                const int c = t / (BN * 8), col = (t / 8) % BN, g = t & 7;
                const int j = col >> 4, cc = col & 15;
                const int gn = n0 + j * a.gstride + cc, cg = n0 + cc + (a.gstride == 16 ? j * 16 : 0);
                const int gk = kr + c * KB + g * 8;
                const bool okr = (t < R * BN * 8) && cg < a.gwidth && gn < a.N;
                const unsigned off = ((unsigned)gn * (unsigned)a.ldb + (unsigned)gk) * 4u;
                vbk[i][0] = ldg16x<HB>(Bb, off, okr && gk < kend, okr && gk + 2 < kend);
                vbk[i][1] = ldg16x<HB>(Bb, off + 16u, okr && gk + 4 < kend, okr && gk + 6 < kend);
            }
        } else {
#pragma unroll
            for (int i = 0; i < TBN_N; ++i) {
                const int t = tid + THREADS * i;
                const int c = t / (BN * 2), q = (t / 8) % (BN / 4), g = t & 7;
                const int col = q * 4, j = col >> 4, cc = col & 15;
                // (LSTM_FWD0: column quad q = gate q of the four units n0 .. n0+3)
                const int gn = QUADF ? n0 + q * a.gstride : n0 + j * a.gstride + cc;
                const int cg = QUADF ? n0 : n0 + cc + (a.gstride == 16 ? j * 16 : 0);
                const int gk = kr + c * KB + g * 8;
                const bool okc = (t < R * BN * 2) && cg < a.gwidth && gn < a.N;
                const bool okh = okc && cg + 2 < a.gwidth && gn + 2 < a.N;     // upper half of the column quad
                const unsigned off = ((unsigned)gk * (unsigned)a.ldb + (unsigned)gn) * 4u;
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    vbn[i][r] = ldg16x<HB>(Bb, off + (unsigned)r * ((unsigned)a.ldb * 4u), okc && gk + r < kend,
                                       okh && gk + r < kend);
            }
        }
    };
    // one uniform branch per round picks the straight-line variant for the operands' alignment
    auto issue_loads = [&](int kr) __attribute__((always_inline)) {
        using T_ = std::true_type; using F_ = std::false_type;
        if (!a8 && !b8) issue_loads_t(kr, F_{}, F_{});
        else if (!a8) issue_loads_t(kr, F_{}, T_{});
        else if (!b8) issue_loads_t(kr, T_{}, F_{});
        else issue_loads_t(kr, T_{}, T_{});
    };
    // (no software pipeline over rounds here: measured A/B it costs the fp32 kernel 15-20 % --
    // its rounds are MFMA-bound already, the early loads only lengthen the register live ranges)
    for (int kr = kbeg; kr < kend; kr += R * KB) {
        if (kr > kbeg) __syncthreads();                                   // images of the previous round consumed
        issue_loads(kr);
        // ---- round to bf16 and store the images
#pragma unroll
        for (int i = 0; i < TA_N; ++i) {
            const int t = tid + THREADS * i;
            const int c = t / (BM * 8), row = (t / 8) % BM, g = t & 7;
            if (t < R * BM * 8) {
                float* dst = &ImgA[(c * BM + row) * KB];
                *reinterpret_cast<float4*>(dst + (((2 * g) ^ (row & 15)) << 2)) = va[i][0];
                *reinterpret_cast<float4*>(dst + (((2 * g + 1) ^ (row & 15)) << 2)) = va[i][1];
            }
        }
        if (TB) {
#pragma unroll
            for (int i = 0; i < TBK_N; ++i) {
                const int t = tid + THREADS * i;
                const int c = t / (BN * 8), col = (t / 8) % BN, g = t & 7;
                if (t < R * BN * 8) {
                    float* dst = &ImgB[(c * BN + col) * KB];
                    *reinterpret_cast<float4*>(dst + (((2 * g) ^ (col & 15)) << 2)) = vbk[i][0];
                    *reinterpret_cast<float4*>(dst + (((2 * g + 1) ^ (col & 15)) << 2)) = vbk[i][1];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TBN_N; ++i) {
                const int t = tid + THREADS * i;
                const int c = t / (BN * 2), q = (t / 8) % (BN / 4), g = t & 7;
                if (t < R * BN * 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int col = q * 4 + e;
                        auto el = [&](int r) { const float4& x = vbn[i][r]; return e == 0 ? x.x : e == 1 ? x.y : e == 2 ? x.z : x.w; };
                        float* dst = &ImgB[(c * BN + col) * KB];
                        *reinterpret_cast<float4*>(dst + (((2 * g) ^ (col & 15)) << 2)) = make_float4(el(0), el(1), el(2), el(3));
                        *reinterpret_cast<float4*>(dst + (((2 * g + 1) ^ (col & 15)) << 2)) = make_float4(el(4), el(5), el(6), el(7));
                    }
                }
            }
        }
        __syncthreads();
        // ---- MFMAs: wave w owns images w, w+4, ... of the round
#pragma unroll
        for (int cc = 0; cc < (R + 3) / 4; ++cc) {
            const int c = wave + 4 * cc;
            if (c < R && kr + c * KB < kend) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    // lane l supplies k = 16 kb + 4 (l >> 4) + e to the e-th of four MFMAs: one 16-byte
                    // read per operand tile feeds four v_mfma_f32_16x16x4_f32 (A and B use the same k map)
                    const int slot = kb * 4 + (lane >> 4);
                    float4 av[TM], bv[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = i * 16 + (lane & 15);
                        av[i] = *reinterpret_cast<const float4*>(&ImgA[(c * BM + row) * KB + ((slot ^ (row & 15)) << 2)]);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = j * 16 + (lane & 15);
                        bv[j] = *reinterpret_cast<const float4*>(&ImgB[(c * BN + col) * KB + ((slot ^ (col & 15)) << 2)]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) {
                                const float ae = e == 0 ? av[i].x : e == 1 ? av[i].y : e == 2 ? av[i].z : av[i].w;
                                const float be = e == 0 ? bv[j].x : e == 1 ? bv[j].y : e == 2 ? bv[j].z : bv[j].w;
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae, be, acc[i][j], 0, 0, 0);
                            }
                }
            }
        }
    }
    __syncthreads();                                                      // Red aliases the images
    reduce_waves<TM, TN>(acc, Red, lane, wave);
    if (nslab > 1) {
        float* Cz = a.C + (size_t)zslab * a.slab_stride;
        for (int t = wave; t < TM * TN; t += 4) {
            const int i = t / TN, j = t % TN;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + i * 16 + (lane >> 4) * 4 + q;
                const int n = n0 + j * 16 + (lane & 15);
                if (m < a.M && n < a.N) Cz[(size_t)m * a.ldc + n] = Red[(t * 4 + q) * 64 + lane];
            }
        }
        return;
    }
    epilogue<TM, TN, EPI_>(a, pre, Red, m0, n0, lane, wave);
}


bool use_bf16_v2(const Args& a, bool ta, bool tb);

template <int TM, int TN, bool TA, bool TB>
int launch(const air_gemm_t* g, const Args& a0, hipStream_t s) {
    Args a = a0;
    constexpr int BM = 16 * TM, BN = 16 * TN;
    const bool grouped = a.gstride != 16;
    const int ncols = grouped ? a.gwidth : a.N;
    const int tile_cols = a.epi == AIR_EPI_LSTM_FWD0 ? 4 : (grouped ? 16 : BN);      // units (grouped) or columns per tile
    dim3 grid((ncols + tile_cols - 1) / tile_cols, (a.M + BM - 1) / BM, 1);
    const int ks = g->ksplit > 1 ? g->ksplit : 1;
    a.kslab = ((a.K + ks - 1) / ks + 3) & ~3;
    if (a.job_on) {
        // enough planes for ~1 quad of noise per thread (the prologue then ends well inside the GEMM)
        const long quads = (a.job.n_normal + 3) / 4 + (a.job.n_uniform + 3) / 4 + (a.job.twin_n + 3) / 4;
        const long plane = (long)grid.x * grid.y * THREADS;
        long planes = (quads + plane - 1) / plane;
        a.job_on = (int)(planes < 1 ? 1 : (planes > 16 ? 16 : planes));
    }
    grid.z = (a.K + a.kslab - 1) / a.kslab + a.job_on;
    a.slab_stride = (long)a.M * a.ldc;
    // bf16 twins of the operands supplied and this (tile, epilogue, layout) exists as a twin kernel
    if (g->precision == 1 && !TA && twin_rounds(a, TM, TN, false, TB) > 0) return twin_launch(a, TM, TN, TB, grid, s);
    // the lean kernels are instantiated per epilogue; a fused epilogue exists for its one tile shape
    // (resolve_tile) -- any other combination would be a dispatch bug
    constexpr bool T14 = TM == 1 && TN == 4, T12 = TM == 1 && TN == 2, T11 = TM == 1 && TN == 1;
    const int epi = a.epi;
    const bool epi_ok = epi == AIR_EPI_GENERIC || (epi == AIR_EPI_LSTM_FWD && T14 && !TB) || (epi == AIR_EPI_REPARAM_FWD && T12 && !TB) ||
                        (epi == AIR_EPI_LSTM_FWD0 && T11 && !TB && !TA) ||
                        ((epi == AIR_EPI_LSTM_BWD || epi == AIR_EPI_LSTM_BWD_TAIL || epi == AIR_EPI_REPARAM_BWD) && T11);
    if (!epi_ok) return AIR_EINVAL;
    // the four-unit column map of LSTM_FWD0 only exists in the lean kernels
    if (epi == AIR_EPI_LSTM_FWD0 && !use_bf16_v2(a, TA, TB)) return AIR_EALIGN;
#define AIR_V2_LAUNCH(KERNEL, LDS)                                                                                     \
    do {                                                                                                                \
        if (epi == AIR_EPI_GENERIC) hipLaunchKernelGGL((KERNEL<TM, TN, TB, AIR_EPI_GENERIC>), grid, dim3(THREADS), LDS, s, a);      \
        else if constexpr (T14 && !TB) hipLaunchKernelGGL((KERNEL<1, 4, false, AIR_EPI_LSTM_FWD>), grid, dim3(THREADS), LDS, s, a);   \
        else if constexpr (T12 && !TB) hipLaunchKernelGGL((KERNEL<1, 2, false, AIR_EPI_REPARAM_FWD>), grid, dim3(THREADS), LDS, s, a); \
        else if constexpr (T11) {                                                                                      \
            if constexpr (!TB) { if (epi == AIR_EPI_LSTM_FWD0) { hipLaunchKernelGGL((KERNEL<1, 1, false, AIR_EPI_LSTM_FWD0>), grid, dim3(THREADS), LDS, s, a); break; } } \
            if (epi == AIR_EPI_LSTM_BWD) hipLaunchKernelGGL((KERNEL<1, 1, TB, AIR_EPI_LSTM_BWD>), grid, dim3(THREADS), LDS, s, a);   \
            else if (epi == AIR_EPI_LSTM_BWD_TAIL) hipLaunchKernelGGL((KERNEL<1, 1, TB, AIR_EPI_LSTM_BWD_TAIL>), grid, dim3(THREADS), LDS, s, a); \
            else hipLaunchKernelGGL((KERNEL<1, 1, TB, AIR_EPI_REPARAM_BWD>), grid, dim3(THREADS), LDS, s, a);             \
        }                                                                                                               \
    } while (0)
    if constexpr (T14 && !TB && !TA) {
        // exact-fp32 LSTM step on four-unit x four-gate tiles (as the bf16-twin path does): 4 x the workgroups, a quarter of
        // the operand bytes each
        if (g->precision == 0 && epi == AIR_EPI_LSTM_FWD && (a.gwidth & 3) == 0 && use_bf16_v2(a, TA, TB) && !a.job_on) {
            using CfgQ = F32V2Cfg<1, 1>;
            auto kq = gemm_f32v2_kernel<1, 1, false, EPI_LSTM_FWD_Q>;
            const int rcq = air_grant_lds(reinterpret_cast<const void*>(kq), CfgQ::BYTES);
            if (rcq) return rcq;
            dim3 gq((a.gwidth + 3) / 4, grid.y, grid.z);
            hipLaunchKernelGGL(kq, gq, dim3(THREADS), CfgQ::BYTES, s, a);
            AIR_CHECK_LAUNCH();
            return 0;
        }
    }
    if (g->precision == 1) {
        const bool v2 = use_bf16_v2(a, TA, TB);
        if (v2) AIR_V2_LAUNCH(gemm_bf16v2_kernel, 0);
        else hipLaunchKernelGGL((gemm_bf16_kernel<TM, TN, TA, TB>), grid, dim3(THREADS), 0, s, a);
    }
    else if (use_bf16_v2(a, TA, TB)) {   // same operand requirements
        using Cfg = F32V2Cfg<TM, TN>;
        if (Cfg::BYTES > 48 * 1024) {
            // opt-in to the large dynamic LDS once per (kernel function, device): all epilogue variants of this tile
            const void* fns[] = {
                reinterpret_cast<const void*>(&gemm_f32v2_kernel<TM, TN, TB, AIR_EPI_GENERIC>),
                T14 && !TB ? reinterpret_cast<const void*>(&gemm_f32v2_kernel<1, 4, false, AIR_EPI_LSTM_FWD>) : nullptr,
                T12 && !TB ? reinterpret_cast<const void*>(&gemm_f32v2_kernel<1, 2, false, AIR_EPI_REPARAM_FWD>) : nullptr,
                T11 ? reinterpret_cast<const void*>(&gemm_f32v2_kernel<1, 1, TB, AIR_EPI_LSTM_BWD>) : nullptr,
                T11 ? reinterpret_cast<const void*>(&gemm_f32v2_kernel<1, 1, TB, AIR_EPI_LSTM_BWD_TAIL>) : nullptr,
                T11 ? reinterpret_cast<const void*>(&gemm_f32v2_kernel<1, 1, TB, AIR_EPI_REPARAM_BWD>) : nullptr,
                T11 && !TB ? reinterpret_cast<const void*>(&gemm_f32v2_kernel<1, 1, false, AIR_EPI_LSTM_FWD0>) : nullptr};
            for (const void* fn : fns)
                if (fn) {
                    const int rc = air_grant_lds(fn, Cfg::BYTES);
                    if (rc) return rc;
                }
        }
        AIR_V2_LAUNCH(gemm_f32v2_kernel, Cfg::BYTES);
    } else
        hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, TA, TB>), grid, dim3(THREADS), 0, s, a);
#undef AIR_V2_LAUNCH
    AIR_CHECK_LAUNCH();
    return 0;
}

void resolve_tile(const air_gemm_t* g, int& tm, int& tn) {
    tm = g->tile_m; tn = g->tile_n;
    if (g->epi == AIR_EPI_LSTM_FWD) { tm = 1; tn = 4; }
    else if (g->epi == AIR_EPI_REPARAM_FWD) { tm = 1; tn = 2; }
    else if (g->epi == AIR_EPI_LSTM_FWD0) { tm = 1; tn = 1; }
    else if (g->epi != AIR_EPI_GENERIC) { tm = 1; tn = 1; }
    if (tm == 0 || tn == 0) {
        // at these sizes wall time ~ one workgroup's latency: prefer many small workgroups.  (Tried: the tile
        // that minimises the operand-panel bytes of the busiest CU -- 16x32 for N = 512, 32x32 for N = 784.
        // Slower: 0.205 -> 0.218 ms per step; the 16x16 tiles win although some CUs then run 2-3 of them.)
        const long t11 = (long)((g->M + 15) / 16) * ((g->N + 15) / 16);
        if (t11 <= 1024) { tm = 1; tn = 1; }
        else if (t11 <= 4096) { tm = 2; tn = 2; }
        else { tm = 2; tn = 4; }
        // (larger tiles for the products beyond 1024 16x16 tiles -- M = N*B = 1280 rows at 128 x 128 -- measured in round 4:
        // 0.570 -> 0.655 .. 0.73 ms per step; these kernels split K over their waves and live on occupancy)
    }
}

// lean bf16 variant: 8-byte alignment and even K / N / group strides suffice (it splits its 16-byte loads)
bool use_bf16_v2(const Args& a, bool ta, bool tb) {
    auto al8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
    return !ta && al8(a.A) && al8(a.B) && (a.lda & 1) == 0 && (a.ldb & 1) == 0 && (a.K & 1) == 0 &&
           (tb || ((a.N & 1) == 0 && (a.gstride & 1) == 0 && (a.gwidth & 1) == 0));
}

template <bool TA, bool TB>
int pick_tile(const air_gemm_t* g, const Args& a, hipStream_t s) {
    int tm, tn;
    resolve_tile(g, tm, tn);
#define AIR_TILE(TM_, TN_) if (tm == TM_ && tn == TN_) return launch<TM_, TN_, TA, TB>(g, a, s)
    AIR_TILE(1, 1); AIR_TILE(1, 2); AIR_TILE(1, 4); AIR_TILE(2, 2); AIR_TILE(2, 4); AIR_TILE(4, 1); AIR_TILE(4, 2); AIR_TILE(4, 4);
#undef AIR_TILE
    return AIR_EINVAL;
}

}  // namespace

extern "C" int air_gemm_slabs(int K, int ksplit) {
    if (K <= 0) return 0;
    const int ks = ksplit > 1 ? ksplit : 1;
    const int kslab = ((K + ks - 1) / ks + 3) & ~3;
    return (K + kslab - 1) / kslab;
}

static int fill_args(const air_gemm_t* g, Args& a);

extern "C" int air_gemm(const air_gemm_t* g, void* stream) {
    Args a;
    const int rc = fill_args(g, a);
    if (rc) return rc;
    hipStream_t s = air_stream(stream);
    if (g->tile_m == 8 && g->tile_n == 4) {
        // the throughput tiling (fp32 A x bf16 shadow, split-K slabs): air_gemm_bf16.hip::gemm_xw_tp_kernel
        const int ks = g->ksplit > 1 ? g->ksplit : 1;
        a.kslab = ((a.K + ks - 1) / ks + 3) & ~3;
        const int ok = xw_tp_ok(a, g->precision, g->transA != 0, g->transB != 0, g->ksplit);
        return ok ? ok : xw_tp_launch(a, 0, s);
    }
    if (g->transA) return pick_tile<true, false>(g, a, s);
    if (g->transB) return pick_tile<false, true>(g, a, s);
    return pick_tile<false, false>(g, a, s);
}

/* name of the kernel function this descriptor dispatches to, as rocprofv3 prints it
 * (profiling tools match per-op timings with the kernel-trace summary by it) */
extern "C" int air_gemm_kernel_name(const air_gemm_t* g, char* buf, int n) {
    Args a;
    const int rc = fill_args(g, a);
    if (rc) return rc;
    if (!buf || n <= 0) return AIR_EINVAL;
    int tm, tn;
    resolve_tile(g, tm, tn);
    const bool ta = g->transA != 0, tb = g->transB != 0;
    {
        const int ks = g->ksplit > 1 ? g->ksplit : 1;
        a.kslab = ((a.K + ks - 1) / ks + 3) & ~3;
    }
    if (g->tile_m == 8 && g->tile_n == 4) {
        const int ok = xw_tp_ok(a, g->precision, ta, tb, g->ksplit);
        if (ok) return ok;
        snprintf(buf, n, "gemm_xw_tp_kernel<%d>", xw_tp_columns(a));
        return 0;
    }
    if (g->precision == 1 && !ta && twin_rounds(a, tm, tn, false, tb) > 0) {
        twin_kernel_name(a, tm, tn, tb, buf, n);
        return 0;
    }
    if (g->precision == 1 && use_bf16_v2(a, ta, tb))
        snprintf(buf, n, "gemm_bf16v2_kernel<%d, %d, %s, %d>", tm, tn, tb ? "true" : "false", g->epi);
    else if (g->precision == 0 && use_bf16_v2(a, ta, tb)) {
        if (g->epi == AIR_EPI_LSTM_FWD && !ta && !tb && (a.gwidth & 3) == 0 && !g->step_job)
            snprintf(buf, n, "gemm_f32v2_kernel<1, 1, false, %d>", EPI_LSTM_FWD_Q);
        else
            snprintf(buf, n, "gemm_f32v2_kernel<%d, %d, %s, %d>", tm, tn, tb ? "true" : "false", g->epi);
    }
    else
        snprintf(buf, n, "gemm_%s_kernel<%d, %d, %s, %s>", g->precision == 1 ? "bf16" : "f32", tm, tn,
                 ta ? "true" : "false", tb ? "true" : "false");
    return 0;
}

static int fill_args(const air_gemm_t* g, Args& a) {
    if (!g || !g->A || !g->B || !g->C) return AIR_EINVAL;
    if (g->M <= 0 || g->N <= 0 || g->K <= 0) return AIR_EINVAL;
    if (g->precision != 0 && g->precision != 1) return AIR_EINVAL;
    if ((g->act == AIR_ACT_SIGMOID_NOISE || g->actgrad != AIR_GRAD_NONE) && !g->aux) return AIR_EINVAL;
    if (g->transA && g->transB) return AIR_EINVAL;     // never needed on this path
    if (g->epi < AIR_EPI_GENERIC || g->epi > AIR_EPI_LSTM_FWD0) return AIR_EINVAL;
    if (g->ksplit > 1 && g->epi != AIR_EPI_GENERIC) return AIR_EINVAL;
    if (g->addend_slabs > 8) return AIR_ELIMIT;
    a.A = g->A; a.B = g->B; a.C = g->C;
    a.M = g->M; a.N = g->N; a.K = g->K; a.lda = g->lda; a.ldb = g->ldb; a.ldc = g->ldc;
    a.gstride = 16; a.gwidth = g->N;
    a.kslab = g->K; a.slab_stride = 0;
    a.bias = g->bias; a.addend = g->addend; a.aux = g->aux;
    a.ldadd = g->ldadd; a.ldaux = g->ldaux;
    a.add_slabs = g->addend ? (g->addend_slabs > 0 ? g->addend_slabs : 1) : 0;
    a.add_slab_stride = (long)g->M * g->ldadd;
    a.aux_scale = g->aux_scale;
    a.act = g->act; a.actgrad = g->actgrad; a.accumulate = g->accumulate; a.epi = g->epi;
    a.p0 = g->p0; a.p1 = g->p1; a.p2 = g->p2; a.p3 = g->p3; a.p4 = nullptr;
    a.q0 = g->q0; a.q1 = g->q1; a.q2 = g->q2; a.q3 = nullptr;
    a.i0 = g->i0; a.i1 = 0;
    a.A16 = g->A16; a.B16 = g->B16; a.C16 = g->C16; a.q0_16 = g->q0_16; a.q2_16 = g->q2_16;
    a.B16p = (g->precision == 1 && !g->transA && !g->transB) ? g->B16p : nullptr;
    a.job_on = 0;
    if (g->step_job) {
        const air_step_job_t& j = *g->step_job;
        if (!j.dyn || !j.istate || j.nsched < 0 || j.nsched > THREADS || (j.nsched > 0 && !j.sched)) return AIR_EINVAL;
        if (j.n_normal < 0 || j.n_uniform < 0 || (j.n_normal > 0 && !j.normals) || (j.n_uniform > 0 && !j.uniforms)) return AIR_EINVAL;
        a.job_on = 1;
        if (j.twin_n < 0 || (j.twin_n > 0 && (!j.twin_src || !j.twin_dst))) return AIR_EINVAL;
        if (j.twin_n > 0 && (!aligned16(j.twin_src) || (reinterpret_cast<uintptr_t>(j.twin_dst) & 7) != 0)) return AIR_EALIGN;
        a.job = AirStepJob{j.sched, j.nsched, j.dyn, j.istate, j.normals, (long)j.n_normal, j.uniforms, (long)j.n_uniform,
                           (uint32_t)(j.seed & 0xffffffffu), (uint32_t)(j.seed >> 32), j.twin_src, j.twin_dst, (long)j.twin_n};
    }
    switch (g->epi) {
        case AIR_EPI_LSTM_FWD:      // N = 4R gate columns, groups of R
            if (g->transA || g->transB || (g->N & 3) || !g->p0 || !g->q0 || !g->q1 || !g->q2) return AIR_EINVAL;
            a.gstride = g->N / 4; a.gwidth = g->N / 4; break;
        case AIR_EPI_LSTM_FWD0:     // N = 4R, tiles of four units x four gates; plain operands, lean kernels only
            if (g->transA || g->transB || (g->N & 15) || g->addend || !g->q0 || !g->q1 || !g->q2) return AIR_EINVAL;
            a.gstride = g->N / 4; a.gwidth = g->N / 4; break;
        case AIR_EPI_REPARAM_FWD:   // N = 2Z (mean | log_var)
            if (g->transA || g->transB || (g->N & 1) || !g->p0 || !g->q0) return AIR_EINVAL;
            a.gstride = g->N / 2; a.gwidth = g->N / 2; break;
        case AIR_EPI_LSTM_BWD:      // N = R
            if (!g->p0 || !g->p1 || !g->p2 || !g->q0 || !g->q1) return AIR_EINVAL;
            a.gwidth = g->N; break;
        case AIR_EPI_LSTM_BWD_TAIL: // N = R, rows >= i0 are the last step
            if (!g->p0 || !g->p1 || !g->p2 || !g->q0 || !g->q1 || g->p3 || g->i0 < 0 || g->i0 > g->M) return AIR_EINVAL;
            a.gwidth = g->N; break;
        case AIR_EPI_REPARAM_BWD:   // N = Z
            if (!g->p0 || !g->p1 || !g->p2 || !g->p3) return AIR_EINVAL;
            a.gwidth = g->N; break;
        default: break;
    }
    return 0;
}
