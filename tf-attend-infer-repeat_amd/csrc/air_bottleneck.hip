// The VAE bottleneck as ONE launch per direction (vae.py:16-34; bf16 MFMA operands, fp32 accumulate).
//
// Forward:  ml = X.Wml + b (mean | log-variance), z = mean + eps*sqrt(exp(lv)), g = softplus(z.Wg + bg)
// Backward: d_z = dG.Wg^T, (d_mean | d_lv) from the reparameterisation + KL terms, d_x = (d_ml.Wml^T) * softplus'(x)
//
// As two GEMM launches each, these are the most latency-bound links of the chain (100 and 50 output
// columns: 4-48 workgroups that still pay a launch and a memory round trip each).  Here a workgroup owns
// 16 rows and a 64-column slice of the second product; the 2Z-wide first product is recomputed by each
// of the H/64 slices (it is tiny), stays in registers, goes through the reparameterisation in the MFMA
// accumulator layout and is handed to the second product through LDS -- no global round trip, no second
// launch.  Every global load of the workgroup is issued before the first is consumed.
//
// Operand images in LDS are [row or column][k] bf16 with a row stride of K + 8 elements: consecutive rows
// shift by 16 bytes, so the 16-lane ds_read_b128 fragment groups are bank-conflict free without a swizzle.
#include "air_common.h"
#include <atomic>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int THREADS = 256;
constexpr int PAD = 8;

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ unsigned short bf16_of(float x) { return (unsigned short)(pack_bf16(x, 0.0f) & 0xffffu); }

// 16-byte load of a row-major operand with the out-of-range case redirected to element 0 (branch-free;
// the caller zeroes what was out of range when it consumes the value)
__device__ __forceinline__ float4 fetch16(const float* __restrict__ base, bool ok, size_t at) {
    return *reinterpret_cast<const float4*>(base + (ok ? at : 0));
}
__device__ __forceinline__ float4 zero_unless(bool ok, float4 v) {
    return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float comp(const float4& t, int j) { return j == 0 ? t.x : j == 1 ? t.y : j == 2 ? t.z : t.w; }

// twin operands: 8 / 16 bytes of bf16 with the out-of-range case redirected to element 0 and zeroed by a bit mask
__device__ __forceinline__ uint2 fetch8h(const unsigned short* __restrict__ base, bool ok, size_t at) {
    const uint2 t = *reinterpret_cast<const uint2*>(base + (ok ? at : 0));
    const unsigned m = ok ? 0xffffffffu : 0u;
    return make_uint2(t.x & m, t.y & m);
}
__device__ __forceinline__ uint4 fetch16h(const unsigned short* __restrict__ base, bool ok, size_t at) {
    const uint4 t = *reinterpret_cast<const uint4*>(base + (ok ? at : 0));
    const unsigned m = ok ? 0xffffffffu : 0u;
    return make_uint4(t.x & m, t.y & m, t.z & m, t.w & m);
}
// element j (0..3) of the four bf16 a row piece holds, from two rows -> one dword of a [column][k] image
__device__ __forceinline__ unsigned pair16(const uint2& r0, const uint2& r1, int j) {
    const unsigned a = j < 2 ? r0.x : r0.y, b = j < 2 ? r1.x : r1.y;
    return (j & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
}

// fragment of a [row][k] image: lane l -> row (l & 15), the 8 k of slot ks*4 + (l >> 4)
__device__ __forceinline__ bf16x8 frag(const unsigned short* img, int ld, int row0, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(&img[(row0 + (lane & 15)) * ld + ks * 32 + (lane >> 4) * 8]);
}

struct FwdArgs {
    const float* X; const float* Wml; const float* bml; const float* eps; const float* Wg; const float* bg;
    float* ml; float* z; float* g;
    int M, Z, H, ldx;
    unsigned short* z16; unsigned short* g16;      // bf16 twins of z / g (nullable)
    const unsigned short* X16; const unsigned short* Wml16; const unsigned short* Wg16;   // bf16 twins of the operands (TW)
    int ldz;                                       // row stride of z / z16 (>= Z)
};

// K1 = width of X (the last recognition layer), compile-time
// TW: X, Wml and Wg are read from their bf16 twins (the producing GEMM's C16 / Adam's shadow): half the bytes through the
// CU's vector-memory path (65 instead of 129 KB per workgroup), no conversion; the RNE twins are what pack_bf16 makes of
// the fp32 arrays, so both forms give the same bits.
template <int K1, bool TW>
__global__ __launch_bounds__(THREADS) void bottleneck_fwd_kernel(FwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    constexpr int L1 = K1 + PAD, L2 = 64 + PAD;
    unsigned short* A1 = smem;                    // [16][L1]   X tile
    unsigned short* B1 = A1 + 16 * L1;            // [128][L1]  Wml columns: mean unit u at u, log-variance unit u at 64 + u
    unsigned short* A2 = B1 + 128 * L1;           // [16][L2]   z tile, k >= Z zero
    unsigned short* B2 = A2 + 16 * L2;            // [64][L2]   Wg columns of this slice, k >= Z zero

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 64;
    const int M = a.M, Z = a.Z, H = a.H, Z2 = 2 * a.Z;

    // ---- every load of the workgroup, back to back ------------------------------------------------
    // X tile: 16 x K1 floats, consecutive lanes along the row
    constexpr int NX = 16 * K1 / 4 / THREADS;
    float4 vx[TW ? 1 : NX];
    constexpr int NXH = 16 * K1 / 8 / THREADS;     // twin: 16-byte pieces of 8 k
    uint4 vxh[TW ? NXH : 1];
    if (TW) {
#pragma unroll
        for (int i = 0; i < NXH; ++i) {
            const int t = tid + THREADS * i, row = t / (K1 / 8), c8 = t % (K1 / 8);
            vxh[i] = fetch16h(a.X16, m0 + row < M, (size_t)(m0 + row) * a.ldx + c8 * 8);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int t = tid + THREADS * i, row = t / (K1 / 4), c4 = t % (K1 / 4);
            vx[i] = fetch16(a.X, m0 + row < M, (size_t)(m0 + row) * a.ldx + c4 * 4);
        }
    }
    // Wml [K1, 2Z] (columns contiguous): task = (k-run g of 8 rows, column quad) -> 8 float4, transposed in registers
    const int NQ = (Z2 + 3) >> 2;
    const int ntask = NQ * (K1 / 8);
    constexpr int NW = 4;                          // tasks per thread: 2Z <= 128 -> <= 32 quads x K1/8 runs <= 1024 (K1 = 256)
    float4 vw[TW ? 1 : NW][8];
    uint2 vwh[TW ? NW : 1][8];
    int wg_[NW], wq_[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int t = tid + THREADS * i;
        const bool ok = t < ntask;
        const int g = ok ? t / NQ : 0, cq = ok ? t - g * NQ : 0;
        wg_[i] = ok ? g : -1; wq_[i] = cq;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (TW) vwh[i][r] = fetch8h(a.Wml16, ok, (size_t)(g * 8 + r) * Z2 + cq * 4);
            else vw[i][r] = fetch16(a.Wml, ok, (size_t)(g * 8 + r) * Z2 + cq * 4);
        }
    }
    // Wg [Z, H] slice (columns n0 .. n0+63 contiguous): threads 0..127, task = (k-run g, column quad)
    float4 vg[TW ? 1 : 8];
    uint2 vgh[TW ? 8 : 1];
    const int gg = (tid >> 4) & 7, gq = tid & 15;
    const bool gcol = tid < 128 && n0 + gq * 4 < H;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (TW) vgh[r] = fetch8h(a.Wg16, gcol && gg * 8 + r < Z, (size_t)(gg * 8 + r) * H + n0 + gq * 4);
        else vg[r] = fetch16(a.Wg, gcol && gg * 8 + r < Z, (size_t)(gg * 8 + r) * H + n0 + gq * 4);
    }
    // epilogue operands in the accumulator layout: row = (lane>>4)*4 + q, unit / column = 16*wave + (lane&15)
    const int u = wave * 16 + (lane & 15);
    const bool uok = u < Z;
    float e_eps[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + (lane >> 4) * 4 + q;
        e_eps[q] = a.eps[(uok && m < M) ? (size_t)m * Z + u : 0];
    }
    const float b_mean = a.bml[uok ? u : 0], b_lv = a.bml[uok ? Z + u : 0];
    const int n = n0 + u;
    const float b_g = a.bg[n < H ? n : 0];

    // ---- into the images (fp32 operands: rounded to bf16 here) -----------------------------------------
    if (TW) {
#pragma unroll
        for (int i = 0; i < NXH; ++i) {
            const int t = tid + THREADS * i, row = t / (K1 / 8), c8 = t % (K1 / 8);
            *reinterpret_cast<uint4*>(&A1[row * L1 + c8 * 8]) = vxh[i];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (wg_[i] < 0) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = wq_[i] * 4 + j;
                if (c >= Z2) continue;
                const int col = c < Z ? c : 64 + (c - Z);
                const uint4 w = make_uint4(pair16(vwh[i][0], vwh[i][1], j), pair16(vwh[i][2], vwh[i][3], j),
                                           pair16(vwh[i][4], vwh[i][5], j), pair16(vwh[i][6], vwh[i][7], j));
                *reinterpret_cast<uint4*>(&B1[col * L1 + wg_[i] * 8]) = w;
            }
        }
        if (tid < 128) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 w = make_uint4(pair16(vgh[0], vgh[1], j), pair16(vgh[2], vgh[3], j),
                                           pair16(vgh[4], vgh[5], j), pair16(vgh[6], vgh[7], j));
                *reinterpret_cast<uint4*>(&B2[(gq * 4 + j) * L2 + gg * 8]) = w;
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int t = tid + THREADS * i, row = t / (K1 / 4), c4 = t % (K1 / 4);
        const float4 x = zero_unless(m0 + row < M, vx[i]);
        uint2 w; w.x = pack_bf16(x.x, x.y); w.y = pack_bf16(x.z, x.w);
        *reinterpret_cast<uint2*>(&A1[row * L1 + c4 * 4]) = w;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        if (wg_[i] < 0) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = wq_[i] * 4 + j;
            if (c >= Z2) continue;
            const int col = c < Z ? c : 64 + (c - Z);
            uint4 w;
            w.x = pack_bf16(comp(vw[i][0], j), comp(vw[i][1], j)); w.y = pack_bf16(comp(vw[i][2], j), comp(vw[i][3], j));
            w.z = pack_bf16(comp(vw[i][4], j), comp(vw[i][5], j)); w.w = pack_bf16(comp(vw[i][6], j), comp(vw[i][7], j));
            *reinterpret_cast<uint4*>(&B1[col * L1 + wg_[i] * 8]) = w;
        }
    }
    if (tid < 128) {
        float4 t8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) t8[r] = zero_unless(gcol && gg * 8 + r < Z, vg[r]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint4 w;
            w.x = pack_bf16(comp(t8[0], j), comp(t8[1], j)); w.y = pack_bf16(comp(t8[2], j), comp(t8[3], j));
            w.z = pack_bf16(comp(t8[4], j), comp(t8[5], j)); w.w = pack_bf16(comp(t8[6], j), comp(t8[7], j));
            *reinterpret_cast<uint4*>(&B2[(gq * 4 + j) * L2 + gg * 8]) = w;
        }
    }
    }
    __syncthreads();

    // ---- first product: wave w owns the units 16w .. 16w+15, mean and log-variance in the same lanes --
    f32x4 am = {0.f, 0.f, 0.f, 0.f}, al = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < K1 / 32; ++ks) {
        const bf16x8 av = frag(A1, L1, 0, ks, lane);
        am = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, frag(B1, L1, wave * 16, ks, lane), am, 0, 0, 0);
        al = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, frag(B1, L1, 64 + wave * 16, ks, lane), al, 0, 0, 0);
    }
    // vae.py:16-24: mean | log_var (+bias), sample = mean + eps*sqrt(exp(lv)); the slice-0 workgroups store them
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (lane >> 4) * 4 + q, m = m0 + row;
        const float mean = am[q] + b_mean, lv = al[q] + b_lv;
        const float zz = mean + e_eps[q] * sqrtf(expf(lv));
        if (blockIdx.y == 0 && uok && m < M) {
            a.ml[(size_t)m * Z2 + u] = mean;
            a.ml[(size_t)m * Z2 + Z + u] = lv;
            a.z[(size_t)m * a.ldz + u] = zz;
            if (a.z16) a.z16[(size_t)m * a.ldz + u] = bf16_of(zz);
        }
        A2[row * L2 + u] = (uok && m < M) ? bf16_of(zz) : (unsigned short)0;
    }
    __syncthreads();

    // ---- second product: wave w owns columns n0 + 16w .. of this slice -------------------------------
    f32x4 ag = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        ag = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(A2, L2, 0, ks, lane), frag(B2, L2, wave * 16, ks, lane), ag, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + (lane >> 4) * 4 + q;
        if (m < M && n < H) {
            const float gv = air_softplus(ag[q] + b_g);
            a.g[(size_t)m * H + n] = gv;
            if (a.g16) a.g16[(size_t)m * H + n] = bf16_of(gv);
        }
    }
}

struct BwdArgs {
    const float* dG; const float* Wg; const float* ml; const float* eps; const float* att; const float* dyn;
    const float* Wml; const float* x;
    float* d_ml; float* d_x;
    int M, Z, K1;
    unsigned short* d_ml16; unsigned short* d_x16; // bf16 twins of d_ml / d_x (nullable)
    const unsigned short* dG16; const unsigned short* Wg16; const unsigned short* Wml16;  // bf16 twins of the operands (TW)
};

// H = width of dG (the first generative layer), compile-time
// TW: dG, Wg and Wml from their bf16 twins -- all three are k-contiguous here, i.e. straight 16 / 8-byte copies into the
// images (46 instead of 93 KB per workgroup through the vector-memory path, no conversion)
template <int H, bool TW>
__global__ __launch_bounds__(THREADS) void bottleneck_bwd_kernel(BwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    constexpr int L1 = H + PAD, L2 = 128 + PAD;
    unsigned short* A1 = smem;                    // [16][L1]   dG tile
    unsigned short* B1 = A1 + 16 * L1;            // [64][L1]   Wg rows (unit z; rows >= Z unused)
    unsigned short* A2 = B1 + 64 * L1;            // [16][L2]   d_ml tile (d_mean at u, d_lv at Z + u), k >= 2Z zero
    unsigned short* B2 = A2 + 16 * L2;            // [64][L2]   Wml rows of this slice, k >= 2Z zero

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 16, j0 = blockIdx.y * 64;
    const int M = a.M, Z = a.Z, Z2 = 2 * a.Z, K1 = a.K1;

    // ---- every load, back to back ---------------------------------------------------------------------
    constexpr int NX = 16 * H / 4 / THREADS;
    constexpr int NXH = 16 * H / 8 / THREADS;
    float4 vx[TW ? 1 : NX];
    uint4 vxh[TW ? NXH : 1];
    if (TW) {
#pragma unroll
        for (int i = 0; i < NXH; ++i) {
            const int t = tid + THREADS * i, row = t / (H / 8), c8 = t % (H / 8);
            vxh[i] = fetch16h(a.dG16, m0 + row < M, (size_t)(m0 + row) * H + c8 * 8);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int t = tid + THREADS * i, row = t / (H / 4), c4 = t % (H / 4);
            vx[i] = fetch16(a.dG, m0 + row < M, (size_t)(m0 + row) * H + c4 * 4);
        }
    }
    // Wg [Z, H]: row z is k-contiguous already
    constexpr int NG = 64 * H / 4 / THREADS;      // covers 64 rows; rows >= Z are not loaded
    constexpr int NGH = 64 * H / 8 / THREADS;
    float4 vg[TW ? 1 : NG];
    uint4 vgh[TW ? NGH : 1];
    if (TW) {
#pragma unroll
        for (int i = 0; i < NGH; ++i) {
            const int t = tid + THREADS * i, row = t / (H / 8), c8 = t % (H / 8);
            vgh[i] = fetch16h(a.Wg16, row < Z, (size_t)row * H + c8 * 8);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int t = tid + THREADS * i, row = t / (H / 4), c4 = t % (H / 4);
            vg[i] = fetch16(a.Wg, row < Z, (size_t)row * H + c4 * 4);
        }
    }
    // Wml [K1, 2Z]: rows j0 .. j0+63, k = 2Z contiguous
    const int NQ = (Z2 + 3) >> 2;                 // float4 per row, <= 32
    const unsigned inv_nq = ((1u << 20) + NQ - 1) / NQ;   // t / NQ == (t * inv_nq) >> 20 for t < 2048, NQ <= 32
    constexpr int NM = 8;                         // 64 rows x <= 32 quads / 256 threads, densely packed: t -> (row t / NQ, quad t % NQ)
    float4 vm[TW ? 1 : NM];
    uint2 vmh[TW ? NM : 1];
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const int t = tid + THREADS * i;
        const int row = (int)(((unsigned)t * inv_nq) >> 20), c4 = t - row * NQ;
        if (TW) vmh[i] = fetch8h(a.Wml16, row < 64 && j0 + row < K1, (size_t)(j0 + row) * Z2 + c4 * 4);
        else vm[i] = fetch16(a.Wml, row < 64 && j0 + row < K1, (size_t)(j0 + row) * Z2 + c4 * 4);
    }
    const int u = wave * 16 + (lane & 15);
    const bool uok = u < Z;
    float e_mean[4], e_lv[4], e_eps[4], e_mask[4], e_x[4];
    const int jn = j0 + u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + (lane >> 4) * 4 + q;
        const bool ok = uok && m < M;
        e_mean[q] = a.ml[ok ? (size_t)m * Z2 + u : 0];
        e_lv[q] = a.ml[ok ? (size_t)m * Z2 + Z + u : 0];
        e_eps[q] = a.eps[ok ? (size_t)m * Z + u : 0];
        e_mask[q] = a.att[m < M ? (size_t)m * AIR_ATT_STRIDE + AIR_ATT_MASK : 0];
        e_x[q] = a.x[(m < M && jn < K1) ? (size_t)m * K1 + jn : 0];
    }
    const float gs = a.dyn[AIR_DYN_GRAD_SCALE], pv = a.dyn[AIR_DYN_VAE_PV], pm = a.dyn[AIR_DYN_VAE_PM];

    // zero the second product's images: their k padding (>= 2Z) must not meet stale LDS contents
    for (int i = tid; i < (16 + 64) * L2 / 8; i += THREADS)
        reinterpret_cast<uint4*>(A2)[i] = make_uint4(0u, 0u, 0u, 0u);

    // ---- into the images (fp32 operands: rounded to bf16 here) --------------------------------------------
    if (TW) {
#pragma unroll
        for (int i = 0; i < NXH; ++i) {
            const int t = tid + THREADS * i, row = t / (H / 8), c8 = t % (H / 8);
            *reinterpret_cast<uint4*>(&A1[row * L1 + c8 * 8]) = vxh[i];
        }
#pragma unroll
        for (int i = 0; i < NGH; ++i) {
            const int t = tid + THREADS * i, row = t / (H / 8), c8 = t % (H / 8);
            *reinterpret_cast<uint4*>(&B1[row * L1 + c8 * 8]) = vgh[i];
        }
    } else {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int t = tid + THREADS * i, row = t / (H / 4), c4 = t % (H / 4);
        const float4 x = zero_unless(m0 + row < M, vx[i]);
        uint2 w; w.x = pack_bf16(x.x, x.y); w.y = pack_bf16(x.z, x.w);
        *reinterpret_cast<uint2*>(&A1[row * L1 + c4 * 4]) = w;
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int t = tid + THREADS * i, row = t / (H / 4), c4 = t % (H / 4);
        const float4 x = zero_unless(row < Z, vg[i]);
        uint2 w; w.x = pack_bf16(x.x, x.y); w.y = pack_bf16(x.z, x.w);
        *reinterpret_cast<uint2*>(&B1[row * L1 + c4 * 4]) = w;
    }
    }
    __syncthreads();                               // the zero fill is complete before anything lands in A2 / B2
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const int t = tid + THREADS * i;
        const int row = (int)(((unsigned)t * inv_nq) >> 20), c4 = t - row * NQ;
        if (row >= 64) continue;
        if (TW) *reinterpret_cast<uint2*>(&B2[row * L2 + c4 * 4]) = vmh[i];
        else {
            const float4 x = zero_unless(j0 + row < K1, vm[i]);       // 2Z % 4 == 0 (Z even): whole quads
            uint2 w; w.x = pack_bf16(x.x, x.y); w.y = pack_bf16(x.z, x.w);
            *reinterpret_cast<uint2*>(&B2[row * L2 + c4 * 4]) = w;
        }
    }

    // ---- first product: d_z for the units of this wave ----------------------------------------------------
    f32x4 az = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < H / 32; ++ks)
        az = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(A1, L1, 0, ks, lane), frag(B1, L1, wave * 16, ks, lane), az, 0, 0, 0);
    // vae.py:22-24 + the KL of air_model.py:386-392: d loss / d mean, d loss / d log-variance
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (lane >> 4) * 4 + q, m = m0 + row;
        const float klg = e_mask[q] * gs;
        const float var = expf(e_lv[q]);
        const float sd = sqrtf(var);
        const float d = az[q];
        const float dmean = d + klg * (e_mean[q] - pm) / pv;
        const float dlv = d * e_eps[q] * 0.5f * sd + klg * 0.5f * (var / pv - 1.0f);
        if (uok && m < M) {
            if (blockIdx.y == 0) {
                a.d_ml[(size_t)m * Z2 + u] = dmean; a.d_ml[(size_t)m * Z2 + Z + u] = dlv;
                if (a.d_ml16) { a.d_ml16[(size_t)m * Z2 + u] = bf16_of(dmean); a.d_ml16[(size_t)m * Z2 + Z + u] = bf16_of(dlv); }
            }
            A2[row * L2 + u] = bf16_of(dmean);
            A2[row * L2 + Z + u] = bf16_of(dlv);
        }
    }
    __syncthreads();

    // ---- second product: d_x = (d_ml . Wml^T) * softplus'(x), columns j0 + 16w .. -----------------------------
    f32x4 ax = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        ax = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(A2, L2, 0, ks, lane), frag(B2, L2, wave * 16, ks, lane), ax, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + (lane >> 4) * 4 + q;
        if (m < M && jn < K1) {
            const float dxv = ax[q] * (1.0f - expf(-e_x[q]));
            a.d_x[(size_t)m * K1 + jn] = dxv;
            if (a.d_x16) a.d_x16[(size_t)m * K1 + jn] = bf16_of(dxv);
        }
    }
}


// ---------------------------------------------------------------------------
// EXACT-fp32 forms of the two kernels (air_bottleneck_*_t.exact_fp32; the fp32 path of the model: the reference's own
// arithmetic).  Same work split and hand-over through LDS as above; operands stay fp32, products run on
// v_mfma_f32_16x16x4_f32.  One 16-byte LDS read of a k-contiguous operand feeds FOUR MFMAs through the k-permutation
// gemm_f32v2_kernel uses: lane l supplies k = 16 kb + 4 (l >> 4) + e to the e-th of them, for A and B alike.  A
// row-major [K, N] weight (Wml and Wg of the forward) is kept AS IT LIES in LDS ([k][N + 4] floats, coalesced copies)
// and its fragments are read as four scalars per 16 k -- a register transposition into [column][k] images would cost
// this latency-bound kernel more than the scalar reads do.  Limits: K1 = 256 / H = 256; forward 2 Z <= 104.
// ---------------------------------------------------------------------------
constexpr int FPAD = 4;                            // floats: 16-byte row shift of the k-contiguous images

template <int K1>
__global__ __launch_bounds__(THREADS) void bottleneck_fwd_f32_kernel(FwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    constexpr int L1 = K1 + FPAD, LW = 104 + FPAD, L2 = 64 + FPAD;
    float* A1 = reinterpret_cast<float*>(smem);   // [16][L1]   X tile, k contiguous
    float* B1 = A1 + 16 * L1;                     // [K1][LW]   Wml as it lies: column c < 2Z of row k at k * LW + c
    float* A2 = B1 + K1 * LW;                     // [16][L2]   z tile, k (unit) contiguous, k >= Z zero
    float* B2 = A2 + 16 * L2;                     // [64][L2]   Wg rows k < Z of this slice as they lie, rows >= Z zero

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 64;
    const int M = a.M, Z = a.Z, H = a.H, Z2 = 2 * a.Z;

    // ---- every load of the workgroup, back to back ------------------------------------------------
    constexpr int NX = 16 * K1 / 4 / THREADS;
    float4 vx[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int t = tid + THREADS * i, row = t / (K1 / 4), c4 = t % (K1 / 4);
        vx[i] = fetch16(a.X, m0 + row < M, (size_t)(m0 + row) * a.ldx + c4 * 4);
    }
    const int NQ = Z2 >> 2;                        // float4 per row of Wml (Z even: whole quads), <= 26
    const int ntask = NQ * K1;
    constexpr int NW = (26 * K1 + THREADS - 1) / THREADS;
    float4 vw[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int t = tid + THREADS * i;
        vw[i] = fetch16(a.Wml, t < ntask, (size_t)t * 4);                  // (rows are contiguous: task t = quad t of the matrix)
    }
    constexpr int NGQ = 64 * 16 / THREADS;         // Wg slice: 64 rows x 16 quads
    float4 vg[NGQ];
#pragma unroll
    for (int i = 0; i < NGQ; ++i) {
        const int t = tid + THREADS * i, row = t >> 4, c4 = t & 15;
        vg[i] = fetch16(a.Wg, row < Z && n0 + c4 * 4 < H, (size_t)row * H + n0 + c4 * 4);
    }
    const int u = wave * 16 + (lane & 15);
    const bool uok = u < Z;
    float e_eps[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + (lane >> 4) * 4 + q;
        e_eps[q] = a.eps[(uok && m < M) ? (size_t)m * Z + u : 0];
    }
    const float b_mean = a.bml[uok ? u : 0], b_lv = a.bml[uok ? Z + u : 0];
    const int n = n0 + u;
    const float b_g = a.bg[n < H ? n : 0];

    // ---- into LDS -------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int t = tid + THREADS * i, row = t / (K1 / 4), c4 = t % (K1 / 4);
        *reinterpret_cast<float4*>(&A1[row * L1 + c4 * 4]) = zero_unless(m0 + row < M, vx[i]);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int t = tid + THREADS * i;
        if (t < ntask) {
            const int row = t / NQ, c4 = t - row * NQ;
            *reinterpret_cast<float4*>(&B1[row * LW + c4 * 4]) = vw[i];
        }
    }
#pragma unroll
    for (int i = 0; i < NGQ; ++i) {
        const int t = tid + THREADS * i, row = t >> 4, c4 = t & 15;
        *reinterpret_cast<float4*>(&B2[row * L2 + c4 * 4]) = zero_unless(row < Z && n0 + c4 * 4 < H, vg[i]);
    }
    __syncthreads();

    // ---- first product: wave w owns the units 16w .. 16w+15, mean and log-variance in the same lanes --
    // (units >= Z read columns that belong to other units or lie in the row padding: finite or not, they only reach
    // output columns nobody stores; the index is clamped so that every read stays inside the image)
    f32x4 am = {0.f, 0.f, 0.f, 0.f}, al = {0.f, 0.f, 0.f, 0.f};
    const int cm = min(u, LW - 1), cl = min(Z + u, LW - 1);
#pragma unroll 4
    for (int kb = 0; kb < K1 / 16; ++kb) {
        const int k0 = kb * 16 + (lane >> 4) * 4;
        const float4 av = *reinterpret_cast<const float4*>(&A1[(lane & 15) * L1 + k0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ae = e == 0 ? av.x : e == 1 ? av.y : e == 2 ? av.z : av.w;
            am = __builtin_amdgcn_mfma_f32_16x16x4f32(ae, B1[(k0 + e) * LW + cm], am, 0, 0, 0);
            al = __builtin_amdgcn_mfma_f32_16x16x4f32(ae, B1[(k0 + e) * LW + cl], al, 0, 0, 0);
        }
    }
    // vae.py:16-24: mean | log_var (+bias), sample = mean + eps*sqrt(exp(lv)); the slice-0 workgroups store them
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (lane >> 4) * 4 + q, m = m0 + row;
        const float mean = am[q] + b_mean, lv = al[q] + b_lv;
        const float zz = mean + e_eps[q] * sqrtf(expf(lv));
        if (blockIdx.y == 0 && uok && m < M) {
            a.ml[(size_t)m * Z2 + u] = mean;
            a.ml[(size_t)m * Z2 + Z + u] = lv;
            a.z[(size_t)m * a.ldz + u] = zz;
        }
        A2[row * L2 + u] = (uok && m < M) ? zz : 0.0f;
    }
    __syncthreads();

    // ---- second product: wave w owns columns n0 + 16w .. of this slice (K = 64 units, zero beyond Z) --
    f32x4 ag = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const int k0 = kb * 16 + (lane >> 4) * 4;
        const float4 av = *reinterpret_cast<const float4*>(&A2[(lane & 15) * L2 + k0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ae = e == 0 ? av.x : e == 1 ? av.y : e == 2 ? av.z : av.w;
            ag = __builtin_amdgcn_mfma_f32_16x16x4f32(ae, B2[(k0 + e) * L2 + u], ag, 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + (lane >> 4) * 4 + q;
        if (m < M && n < H) a.g[(size_t)m * H + n] = air_softplus(ag[q] + b_g);
    }
}

template <int H>
__global__ __launch_bounds__(THREADS) void bottleneck_bwd_f32_kernel(BwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    constexpr int L1 = H + FPAD, L2 = 128 + FPAD;
    float* A1 = reinterpret_cast<float*>(smem);   // [16][L1]   dG tile
    float* B1 = A1 + 16 * L1;                     // [64][L1]   Wg rows (unit z; rows >= Z zero)
    float* A2 = B1 + 64 * L1;                     // [16][L2]   d_ml tile (d_mean at u, d_lv at Z + u), k >= 2Z zero
    float* B2 = A2 + 16 * L2;                     // [64][L2]   Wml rows of this slice, k >= 2Z zero

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 16, j0 = blockIdx.y * 64;
    const int M = a.M, Z = a.Z, Z2 = 2 * a.Z, K1 = a.K1;

    constexpr int NX = 16 * H / 4 / THREADS;
    float4 vx[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int t = tid + THREADS * i, row = t / (H / 4), c4 = t % (H / 4);
        vx[i] = fetch16(a.dG, m0 + row < M, (size_t)(m0 + row) * H + c4 * 4);
    }
    constexpr int NG = 64 * H / 4 / THREADS;
    float4 vg[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int t = tid + THREADS * i, row = t / (H / 4), c4 = t % (H / 4);
        vg[i] = fetch16(a.Wg, row < Z, (size_t)row * H + c4 * 4);
    }
    const int NQ = (Z2 + 3) >> 2;
    const unsigned inv_nq = ((1u << 20) + NQ - 1) / NQ;
    constexpr int NM = 8;
    float4 vm[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const int t = tid + THREADS * i;
        const int row = (int)(((unsigned)t * inv_nq) >> 20), c4 = t - row * NQ;
        vm[i] = fetch16(a.Wml, row < 64 && j0 + row < K1, (size_t)(j0 + row) * Z2 + c4 * 4);
    }
    const int u = wave * 16 + (lane & 15);
    const bool uok = u < Z;
    float e_mean[4], e_lv[4], e_eps[4], e_mask[4], e_x[4];
    const int jn = j0 + u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + (lane >> 4) * 4 + q;
        const bool ok = uok && m < M;
        e_mean[q] = a.ml[ok ? (size_t)m * Z2 + u : 0];
        e_lv[q] = a.ml[ok ? (size_t)m * Z2 + Z + u : 0];
        e_eps[q] = a.eps[ok ? (size_t)m * Z + u : 0];
        e_mask[q] = a.att[m < M ? (size_t)m * AIR_ATT_STRIDE + AIR_ATT_MASK : 0];
        e_x[q] = a.x[(m < M && jn < K1) ? (size_t)m * K1 + jn : 0];
    }
    const float gs = a.dyn[AIR_DYN_GRAD_SCALE], pv = a.dyn[AIR_DYN_VAE_PV], pm = a.dyn[AIR_DYN_VAE_PM];

    // zero the second product's images: their k padding (>= 2Z) must not meet stale LDS contents
    for (int i = tid; i < (16 + 64) * L2 / 4; i += THREADS)
        reinterpret_cast<float4*>(A2)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int t = tid + THREADS * i, row = t / (H / 4), c4 = t % (H / 4);
        *reinterpret_cast<float4*>(&A1[row * L1 + c4 * 4]) = zero_unless(m0 + row < M, vx[i]);
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int t = tid + THREADS * i, row = t / (H / 4), c4 = t % (H / 4);
        *reinterpret_cast<float4*>(&B1[row * L1 + c4 * 4]) = zero_unless(row < Z, vg[i]);
    }
    __syncthreads();                               // the zero fill is complete before anything lands in A2 / B2
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const int t = tid + THREADS * i;
        const int row = (int)(((unsigned)t * inv_nq) >> 20), c4 = t - row * NQ;
        if (row >= 64) continue;
        *reinterpret_cast<float4*>(&B2[row * L2 + c4 * 4]) = zero_unless(j0 + row < K1, vm[i]);   // 2Z % 4 == 0 (Z even): whole quads
    }

    // ---- first product: d_z for the units of this wave (both operands k-contiguous) ------------------------
    f32x4 az = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int kb = 0; kb < H / 16; ++kb) {
        const int k0 = kb * 16 + (lane >> 4) * 4;
        const float4 av = *reinterpret_cast<const float4*>(&A1[(lane & 15) * L1 + k0]);
        const float4 bv = *reinterpret_cast<const float4*>(&B1[u * L1 + k0]);
        az = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, az, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, az, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, az, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, az, 0, 0, 0);
    }
    // vae.py:22-24 + the KL of air_model.py:386-392: d loss / d mean, d loss / d log-variance
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (lane >> 4) * 4 + q, m = m0 + row;
        const float klg = e_mask[q] * gs;
        const float var = expf(e_lv[q]);
        const float sd = sqrtf(var);
        const float d = az[q];
        const float dmean = d + klg * (e_mean[q] - pm) / pv;
        const float dlv = d * e_eps[q] * 0.5f * sd + klg * 0.5f * (var / pv - 1.0f);
        if (uok && m < M) {
            if (blockIdx.y == 0) { a.d_ml[(size_t)m * Z2 + u] = dmean; a.d_ml[(size_t)m * Z2 + Z + u] = dlv; }
            A2[row * L2 + u] = dmean;
            A2[row * L2 + Z + u] = dlv;
        }
    }
    __syncthreads();

    // ---- second product: d_x = (d_ml . Wml^T) * softplus'(x), columns j0 + 16w .. (K = 128, zero beyond 2Z) ----
    f32x4 ax = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const int k0 = kb * 16 + (lane >> 4) * 4;
        const float4 av = *reinterpret_cast<const float4*>(&A2[(lane & 15) * L2 + k0]);
        const float4 bv = *reinterpret_cast<const float4*>(&B2[u * L2 + k0]);
        ax = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, ax, 0, 0, 0);
        ax = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, ax, 0, 0, 0);
        ax = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, ax, 0, 0, 0);
        ax = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, ax, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + (lane >> 4) * 4 + q;
        if (m < M && jn < K1) a.d_x[(size_t)m * K1 + jn] = ax[q] * (1.0f - expf(-e_x[q]));
    }
}

template <typename K>
int grant_lds(K kernel, size_t bytes) { return air_grant_lds(reinterpret_cast<const void*>(kernel), bytes); }

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

}  // namespace

extern "C" int air_vae_bottleneck_fwd(const air_bottleneck_fwd_t* a, void* stream) {
    if (!a || !a->X || !a->Wml || !a->bml || !a->eps || !a->Wg || !a->bg || !a->ml || !a->z || !a->g) return AIR_EINVAL;
    if (a->M <= 0 || a->K1 <= 0 || a->Z <= 0 || a->H <= 0 || a->ldx < a->K1) return AIR_EINVAL;
    if (a->K1 != 256 || a->Z > 64) return AIR_ELIMIT;
    if ((a->Z & 1) || (a->H & 3) || (a->ldx & 3) || !al16(a->X) || !al16(a->Wml) || !al16(a->Wg)) return AIR_EALIGN;
    if (a->ldz != 0 && a->ldz < a->Z) return AIR_EINVAL;
    const int ldz = a->ldz > 0 ? a->ldz : a->Z;
    constexpr int K1 = 256;
    if (a->exact_fp32) {
        if (2 * a->Z > 104) return AIR_ELIMIT;                      // Wml as it lies: rows of at most 104 + 4 floats of LDS
        const size_t ldsf = sizeof(float) * (16 * (K1 + FPAD) + K1 * (104 + FPAD) + (16 + 64) * (64 + FPAD));
        const int rcf = grant_lds(bottleneck_fwd_f32_kernel<K1>, ldsf);
        if (rcf) return rcf;
        FwdArgs kf{a->X, a->Wml, a->bml, a->eps, a->Wg, a->bg, a->ml, a->z, a->g, a->M, a->Z, a->H, a->ldx, nullptr, nullptr,
                   nullptr, nullptr, nullptr, ldz};
        hipLaunchKernelGGL((bottleneck_fwd_f32_kernel<K1>), dim3((a->M + 15) / 16, (a->H + 63) / 64), dim3(THREADS), ldsf,
                           air_stream(stream), kf);
        AIR_CHECK_LAUNCH();
        return 0;
    }
    const size_t lds = sizeof(unsigned short) * ((16 + 128) * (K1 + PAD) + (16 + 64) * (64 + PAD));
    // twin operands: all three or none; whole 16-byte pieces of X rows, 8-byte pieces of the weight rows
    const bool tw = a->X16 && a->Wml16 && a->Wg16 && al16(a->X16) && (a->ldx & 7) == 0 && al8(a->Wml16) && al8(a->Wg16);
    const int rc = tw ? grant_lds(bottleneck_fwd_kernel<K1, true>, lds) : grant_lds(bottleneck_fwd_kernel<K1, false>, lds);
    if (rc) return rc;
    FwdArgs k{a->X, a->Wml, a->bml, a->eps, a->Wg, a->bg, a->ml, a->z, a->g, a->M, a->Z, a->H, a->ldx, a->z16, a->g16,
              a->X16, a->Wml16, a->Wg16, ldz};
    const dim3 grid((a->M + 15) / 16, (a->H + 63) / 64);
    if (tw) hipLaunchKernelGGL((bottleneck_fwd_kernel<K1, true>), grid, dim3(THREADS), lds, air_stream(stream), k);
    else hipLaunchKernelGGL((bottleneck_fwd_kernel<K1, false>), grid, dim3(THREADS), lds, air_stream(stream), k);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_vae_bottleneck_bwd(const air_bottleneck_bwd_t* a, void* stream) {
    if (!a || !a->dG || !a->Wg || !a->ml || !a->eps || !a->att || !a->dyn || !a->Wml || !a->x || !a->d_ml || !a->d_x) return AIR_EINVAL;
    if (a->M <= 0 || a->K1 <= 0 || a->Z <= 0 || a->H <= 0) return AIR_EINVAL;
    if (a->H != 256 || a->Z > 64) return AIR_ELIMIT;
    if ((a->Z & 1) || !al16(a->dG) || !al16(a->Wg) || !al16(a->Wml)) return AIR_EALIGN;
    constexpr int H = 256;
    if (a->exact_fp32) {
        const size_t ldsf = sizeof(float) * ((16 + 64) * (H + FPAD) + (16 + 64) * (128 + FPAD));
        const int rcf = grant_lds(bottleneck_bwd_f32_kernel<H>, ldsf);
        if (rcf) return rcf;
        BwdArgs kf{a->dG, a->Wg, a->ml, a->eps, a->att, a->dyn, a->Wml, a->x, a->d_ml, a->d_x, a->M, a->Z, a->K1, nullptr, nullptr,
                   nullptr, nullptr, nullptr};
        hipLaunchKernelGGL((bottleneck_bwd_f32_kernel<H>), dim3((a->M + 15) / 16, (a->K1 + 63) / 64), dim3(THREADS), ldsf,
                           air_stream(stream), kf);
        AIR_CHECK_LAUNCH();
        return 0;
    }
    const size_t lds = sizeof(unsigned short) * ((16 + 64) * (H + PAD) + (16 + 64) * (128 + PAD));
    const bool tw = a->dG16 && a->Wg16 && a->Wml16 && al16(a->dG16) && al16(a->Wg16) && al8(a->Wml16);
    const int rc = tw ? grant_lds(bottleneck_bwd_kernel<H, true>, lds) : grant_lds(bottleneck_bwd_kernel<H, false>, lds);
    if (rc) return rc;
    BwdArgs k{a->dG, a->Wg, a->ml, a->eps, a->att, a->dyn, a->Wml, a->x, a->d_ml, a->d_x, a->M, a->Z, a->K1, a->d_ml16, a->d_x16,
              a->dG16, a->Wg16, a->Wml16};
    const dim3 grid((a->M + 15) / 16, (a->K1 + 63) / 64);
    if (tw) hipLaunchKernelGGL((bottleneck_bwd_kernel<H, true>), grid, dim3(THREADS), lds, air_stream(stream), k);
    else hipLaunchKernelGGL((bottleneck_bwd_kernel<H, false>), grid, dim3(THREADS), lds, air_stream(stream), k);
    AIR_CHECK_LAUNCH();
    return 0;
}
