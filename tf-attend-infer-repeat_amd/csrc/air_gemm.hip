// Small-batch GEMM with fused epilogues for the AIR loop (LSTM / heads / VAE
// MatMul + BiasAdd + activation and their gradients).
//
// Shapes on this path are skinny: M = batch (64..256) for forward / data-grad,
// or the contraction is N_steps*batch (192) for weight-grad.  The kernel is
// therefore built for LATENCY, not for peak MFMA rate:
//   * one workgroup = 4 waves = one (16*TM x 16*TN) output tile; the 4 waves
//     split every K-chunk four ways (one wave per SIMD -> 4 matrix pipes work on
//     the same tile) and are summed through LDS in a fixed order (deterministic);
//   * operands are read from HBM/L2 exactly once per workgroup with coalesced
//     loads, register-prefetched one chunk ahead, and transposed through LDS
//     into the MFMA fragment order so all four layout cases (NN, NT, TN) are
//     conflict-free on the read side;
//   * precision 0 uses v_mfma_f32_16x16x4_f32 (exact fp32: bit-equal to an fmaf
//     chain) -- the parity path; precision 1 rounds both operands to bf16 while
//     staging and uses v_mfma_f32_16x16x32_bf16 with fp32 accumulation.
#include "air_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 128;        // K-chunk per workgroup iteration (32 per wave)
constexpr int THREADS = 256;

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

struct Epilogue {
    const float* bias; const float* addend; const float* aux;
    float* C;
    int ldc, ldadd, ldaux;
    float aux_scale;
    int act, actgrad, accumulate;

    __device__ __forceinline__ void apply(float v, int m, int n) const {
        if (bias) v += bias[n];
        if (addend) v += addend[(size_t)m * ldadd + n];
        if (act == AIR_ACT_RELU) v = fmaxf(v, 0.0f);
        else if (act == AIR_ACT_SOFTPLUS) v = air_softplus(v);
        else if (act == AIR_ACT_SIGMOID_NOISE) v = air_sigmoid(v + aux[(size_t)m * ldaux + n] * aux_scale);
        if (actgrad == AIR_GRAD_RELU) v = (aux[(size_t)m * ldaux + n] > 0.0f) ? v : 0.0f;
        else if (actgrad == AIR_GRAD_SOFTPLUS) v = v * (1.0f - expf(-aux[(size_t)m * ldaux + n]));
        float* c = C + (size_t)m * ldc + n;
        if (accumulate) v += *c;
        *c = v;
    }
};

// ---------------------------------------------------------------------------
// fp32 path: v_mfma_f32_16x16x4_f32.  A frag: lane l holds A[m=l&15][k=l>>4];
// B frag: B[k=l>>4][n=l&15]; C/D: col = l&15, row = (l>>4)*4 + reg.
// LDS images are k-major ([k][m] / [k][n]) so a fragment read is 16 consecutive
// floats per k row -> conflict-free.
// ---------------------------------------------------------------------------
template <int TM, int TN, bool TA, bool TB>
__global__ __launch_bounds__(THREADS) void gemm_f32_kernel(
    const float* __restrict__ A, const float* __restrict__ B,
    int M, int N, int K, int lda, int ldb, Epilogue ep)
{
    constexpr int BM = 16 * TM, BN = 16 * TN;
    constexpr int LA = BM + 1, LB = BN + 1;          // +1 pad: transposing stores spread over banks
    constexpr int NA = BM * BK / THREADS;            // staged floats per thread
    constexpr int NB = BN * BK / THREADS;
    __shared__ float As[BK * LA];
    __shared__ float Bs[BK * LB];
    __shared__ float Red[3 * TM * TN * 4 * 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    float ra[NA], rb[NB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * THREADS;
            int m, k;
            if (TA) { k = idx / BM; m = idx % BM; } else { m = idx / BK; k = idx % BK; }
            const int gm = m0 + m, gk = k0 + k;
            float v = 0.0f;
            if (gm < M && gk < K) v = TA ? A[(size_t)gk * lda + gm] : A[(size_t)gm * lda + gk];
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + i * THREADS;
            int n, k;
            if (TB) { n = idx / BK; k = idx % BK; } else { k = idx / BN; n = idx % BN; }
            const int gn = n0 + n, gk = k0 + k;
            float v = 0.0f;
            if (gn < N && gk < K) v = TB ? B[(size_t)gn * ldb + gk] : B[(size_t)gk * ldb + gn];
            rb[i] = v;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * THREADS;
            int m, k;
            if (TA) { k = idx / BM; m = idx % BM; } else { m = idx / BK; k = idx % BK; }
            As[k * LA + m] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + i * THREADS;
            int n, k;
            if (TB) { n = idx / BK; k = idx % BK; } else { k = idx / BN; n = idx % BN; }
            Bs[k * LB + n] = rb[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();                 // previous chunk fully consumed
        stage();
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK); // next chunk in flight under the MFMAs
        const int kw = wave * (BK / 4);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int kk = kw + ks * 4 + (lane >> 4);
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[kk * LA + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[kk * LB + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // cross-wave (split-K) reduction in a fixed order: ((w0 + w1) + w2) + w3
    if (wave > 0) {
        float* r = Red + (wave - 1) * (TM * TN * 4 * 64);
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
                    float v = acc[i][j][q];
#pragma unroll
                    for (int w = 0; w < 3; ++w) v += Red[w * (TM * TN * 4 * 64) + ((i * TN + j) * 4 + q) * 64 + lane];
                    Red[((i * TN + j) * 4 + q) * 64 + lane] = v;   // own slot of region 0: no hazard
                }
    }
    __syncthreads();
    // epilogue by all 4 waves: tile t handled by wave (t & 3)
    for (int t = wave; t < TM * TN; t += 4) {
        const int i = t / TN, j = t % TN;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + i * 16 + (lane >> 4) * 4 + q;
            const int n = n0 + j * 16 + (lane & 15);
            if (m < M && n < N) ep.apply(Red[(t * 4 + q) * 64 + lane], m, n);
        }
    }
}

// ---------------------------------------------------------------------------
// bf16 path: v_mfma_f32_16x16x32_bf16.  A frag: lane l holds 8 consecutive k
// (k = (l>>4)*8 .. +7) of row m = l&15; B frag likewise for column n = l&15.
// LDS images are [m][k] / [n][k] with k contiguous (one 16-byte read per
// fragment); rows padded by 8 halves so the 16 rows of a lane group start on
// different 16-byte slots.
// ---------------------------------------------------------------------------
template <int TM, int TN, bool TA, bool TB>
__global__ __launch_bounds__(THREADS) void gemm_bf16_kernel(
    const float* __restrict__ A, const float* __restrict__ B,
    int M, int N, int K, int lda, int ldb, Epilogue ep)
{
    constexpr int BM = 16 * TM, BN = 16 * TN;
    constexpr int LK = BK + 8;                       // halves per row
    constexpr int NA = BM * BK / THREADS;
    constexpr int NB = BN * BK / THREADS;
    __shared__ __attribute__((aligned(16))) unsigned short As[BM * LK];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[BN * LK];
    __shared__ float Red[3 * TM * TN * 4 * 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    float ra[NA], rb[NB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * THREADS;
            int m, k;
            if (TA) { k = idx / BM; m = idx % BM; } else { m = idx / BK; k = idx % BK; }
            const int gm = m0 + m, gk = k0 + k;
            float v = 0.0f;
            if (gm < M && gk < K) v = TA ? A[(size_t)gk * lda + gm] : A[(size_t)gm * lda + gk];
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + i * THREADS;
            int n, k;
            if (TB) { n = idx / BK; k = idx % BK; } else { k = idx / BN; n = idx % BN; }
            const int gn = n0 + n, gk = k0 + k;
            float v = 0.0f;
            if (gn < N && gk < K) v = TB ? B[(size_t)gn * ldb + gk] : B[(size_t)gk * ldb + gn];
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

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
        stage();
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK);
        // wave w owns k in [w*32, w*32+32) of the chunk: exactly one 16x16x32 step
        const int kk = wave * 32 + (lane >> 4) * 8;
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(&As[(i * 16 + (lane & 15)) * LK + kk]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(&Bs[(j * 16 + (lane & 15)) * LK + kk]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }

    if (wave > 0) {
        float* r = Red + (wave - 1) * (TM * TN * 4 * 64);
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
                    float v = acc[i][j][q];
#pragma unroll
                    for (int w = 0; w < 3; ++w) v += Red[w * (TM * TN * 4 * 64) + ((i * TN + j) * 4 + q) * 64 + lane];
                    Red[((i * TN + j) * 4 + q) * 64 + lane] = v;
                }
    }
    __syncthreads();
    for (int t = wave; t < TM * TN; t += 4) {
        const int i = t / TN, j = t % TN;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + i * 16 + (lane >> 4) * 4 + q;
            const int n = n0 + j * 16 + (lane & 15);
            if (m < M && n < N) ep.apply(Red[(t * 4 + q) * 64 + lane], m, n);
        }
    }
}

template <int TM, int TN, bool TA, bool TB>
int launch(const air_gemm_t* g, const Epilogue& ep, hipStream_t s) {
    dim3 grid((g->N + 16 * TN - 1) / (16 * TN), (g->M + 16 * TM - 1) / (16 * TM));
    if (g->precision == 1)
        hipLaunchKernelGGL((gemm_bf16_kernel<TM, TN, TA, TB>), grid, dim3(THREADS), 0, s,
                           g->A, g->B, g->M, g->N, g->K, g->lda, g->ldb, ep);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, TA, TB>), grid, dim3(THREADS), 0, s,
                           g->A, g->B, g->M, g->N, g->K, g->lda, g->ldb, ep);
    AIR_CHECK_LAUNCH();
    return 0;
}

template <bool TA, bool TB>
int pick_tile(const air_gemm_t* g, const Epilogue& ep, hipStream_t s) {
    // smallest tile that still yields >= ~2 workgroups per CU, else 16x16:
    // at these sizes wall time is one workgroup's latency, so prefer many small ones.
    const long t11 = (long)((g->M + 15) / 16) * ((g->N + 15) / 16);
    if (t11 <= 1024) return launch<1, 1, TA, TB>(g, ep, s);
    if (t11 <= 4096) return launch<2, 2, TA, TB>(g, ep, s);
    return launch<2, 4, TA, TB>(g, ep, s);
}

}  // namespace

extern "C" int air_gemm(const air_gemm_t* g, void* stream) {
    if (!g || !g->A || !g->B || !g->C) return AIR_EINVAL;
    if (g->M <= 0 || g->N <= 0 || g->K <= 0) return AIR_EINVAL;
    if (g->precision != 0 && g->precision != 1) return AIR_EINVAL;
    if ((g->act == AIR_ACT_SIGMOID_NOISE || g->actgrad != AIR_GRAD_NONE) && !g->aux) return AIR_EINVAL;
    if (g->transA && g->transB) return AIR_EINVAL;     // never needed on this path
    Epilogue ep;
    ep.bias = g->bias; ep.addend = g->addend; ep.aux = g->aux; ep.C = g->C;
    ep.ldc = g->ldc; ep.ldadd = g->ldadd; ep.ldaux = g->ldaux; ep.aux_scale = g->aux_scale;
    ep.act = g->act; ep.actgrad = g->actgrad; ep.accumulate = g->accumulate;
    hipStream_t s = air_stream(stream);
    if (g->transA) return pick_tile<true, false>(g, ep, s);
    if (g->transB) return pick_tile<false, true>(g, ep, s);
    return pick_tile<false, false>(g, ep, s);
}
