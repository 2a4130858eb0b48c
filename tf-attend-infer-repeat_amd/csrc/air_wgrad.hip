// Grouped weight-gradient launch: every dW = A^T . dY of the train step (the
// MatMul_grad/BiasAdd_grad nodes of all 36 variables) in ONE kernel.
//
// Weights are shared across the N time steps, so each dW contracts over all
// N*B (step, image) rows: K is tiny (64..192) while M x N is the size of the
// weight matrix -- an outer-product-shaped GEMM.  A 64 x 64 output tile is owned
// by one workgroup; its 4 waves own disjoint 32 x 32 quadrants and run the full
// K loop themselves (no split-K, no cross-wave reduction).  Operand chunks of
// 32 rows are staged through double-buffered LDS with the next chunk's 16-byte
// loads in flight under the MFMAs.  The bias gradient (column sums of dY) is
// taken from the dY chunk already in LDS by the tiles of block-row 0, so no
// separate reduction launch exists.  The ten problems of the step are
// independent: one launch fills the chip instead of ten latency-bound ones.
#include "air_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 256;
constexpr int BT = 64;          // output tile (both dims)
constexpr int KC = 32;          // rows per staged chunk
constexpr int LS = BT + 16;     // LDS row stride: = 16 (mod 32) dwords -> conflict-free fragment reads, 16-B aligned
constexpr int MAXP = 12;

struct Prob {
    const float* A; const float* dY; float* dW; float* db;
    int M, N, K, lda, ldb, ldc;
    int head_pack, Hs, Hh, Hz;
    int tiles_n, first_block;
};
struct Table { int count; int total_blocks; Prob p[MAXP]; };

__device__ __forceinline__ float4 load4(const float* base, int ld, int row, int col, int rows, int cols, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows) {
        const float* src = base + (size_t)row * ld + col;
        if (vec && col + 3 < cols) v = *reinterpret_cast<const float4*>(src);
        else {
            if (col < cols) v.x = src[0];
            if (col + 1 < cols) v.y = src[1];
            if (col + 2 < cols) v.z = src[2];
            if (col + 3 < cols) v.w = src[3];
        }
    }
    return v;
}

__global__ __launch_bounds__(THREADS) void wgrad_grouped_kernel(Table tab)
{
    __shared__ __attribute__((aligned(16))) float As[2][KC * LS];
    __shared__ __attribute__((aligned(16))) float Bs[2][KC * LS];

    // which problem / tile is this workgroup?
    int pi = 0;
    while (pi + 1 < tab.count && (int)blockIdx.x >= tab.p[pi + 1].first_block) ++pi;
    const Prob& pr = tab.p[pi];
    const int local = blockIdx.x - pr.first_block;
    const int m0 = (local / pr.tiles_n) * BT, n0 = (local % pr.tiles_n) * BT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;        // this wave's quadrant
    const bool vecA = ((pr.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(pr.A) & 15) == 0);
    const bool vecB = ((pr.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(pr.dY) & 15) == 0);

    // staging map: 32 rows x 16 float4 columns = 512 float4 per operand chunk, 2 per thread
    const int srow = tid >> 4, scol = (tid & 15) * 4;             // rows srow and srow + 16
    float4 ra[2], rb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + srow + 16 * h;
            ra[h] = load4(pr.A, pr.lda, k, m0 + scol, pr.K, pr.M, vecA);
            rb[h] = load4(pr.dY, pr.ldb, k, n0 + scol, pr.K, pr.N, vecB);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<float4*>(&As[buf][(srow + 16 * h) * LS + scol]) = ra[h];
            *reinterpret_cast<float4*>(&Bs[buf][(srow + 16 * h) * LS + scol]) = rb[h];
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float colsum = 0.0f;                                          // wave 0 of block-row 0: column n0 + lane
    const bool do_bias = (pr.db != nullptr) && (m0 == 0) && (wave == 0);

    const int nchunks = (pr.K + KC - 1) / KC;
    fetch(0);
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        stage(buf);
        __syncthreads();                       // chunk c visible; everyone is past the MFMAs of chunk c-1
        if (c + 1 < nchunks) fetch((c + 1) * KC);
        const float* as = As[buf];
        const float* bs = Bs[buf];
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            const int kk = ks * 4 + (lane >> 4);
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = as[kk * LS + wm + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = bs[kk * LS + wn + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (do_bias) {
#pragma unroll 8
            for (int k = 0; k < KC; ++k) colsum += bs[k * LS + lane];      // zero rows beyond K
        }
        // the stage() of chunk c+1 writes the OTHER buffer; the barrier of iteration c+1 orders it
        // after every wave's reads of chunk c-1 ... and buffer `buf` is rewritten only at c+2
    }

    // epilogue: C/D map row = (lane>>4)*4 + q, col = lane&15
    if (!pr.head_pack) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + wm + i * 16 + (lane >> 4) * 4 + q;
                    const int n = n0 + wn + j * 16 + (lane & 15);
                    if (m < pr.M && n < pr.N) pr.dW[(size_t)m * pr.ldc + n] = acc[i][j][q];
                }
        if (do_bias && n0 + lane < pr.N) pr.db[n0 + lane] = colsum;
    } else {
        // head output units (air_model.py:294-316, 376): A = d_out7 [K,8], dY = hid [K,HT];
        // unit o only owns the hidden segment of its head: dW = wout[o][n - off], db = bout[o] = sum_k d_out7[k][o]
        const int wid[5] = {pr.Hs, pr.Hs, pr.Hh, pr.Hh, pr.Hz};
        const int head[7] = {0, 1, 2, 2, 3, 3, 4};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = m0 + wm + i * 16 + (lane >> 4) * 4 + q;
                    const int n = n0 + wn + j * 16 + (lane & 15);
                    if (o < 7 && n < pr.N) {
                        int off = 0;
                        for (int h = 0; h < head[o]; ++h) off += wid[h];
                        if (n >= off && n < off + wid[head[o]]) pr.dW[(size_t)o * pr.ldc + (n - off)] = acc[i][j][q];
                    }
                }
        if (pr.db != nullptr && n0 == 0 && wave == 0 && lane < 7) {
            float s = 0.0f;
            for (int k = 0; k < pr.K; ++k) s += pr.A[(size_t)k * pr.lda + lane];
            pr.db[lane] = s;
        }
    }
}

}  // namespace

extern "C" int air_wgrad_grouped(const air_wgrad_t* probs, int count, void* stream) {
    if (!probs || count <= 0) return AIR_EINVAL;
    if (count > MAXP) return AIR_ELIMIT;
    Table tab;
    tab.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const air_wgrad_t& g = probs[i];
        if (!g.A || !g.dY || !g.dW || g.M <= 0 || g.N <= 0 || g.K <= 0) return AIR_EINVAL;
        Prob& p = tab.p[i];
        p.A = g.A; p.dY = g.dY; p.dW = g.dW; p.db = g.db;
        p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc;
        p.head_pack = g.head_pack; p.Hs = g.Hs; p.Hh = g.Hh; p.Hz = g.Hz;
        p.tiles_n = (g.N + BT - 1) / BT;
        p.first_block = blocks;
        blocks += p.tiles_n * ((g.M + BT - 1) / BT);
    }
    for (int i = count; i < MAXP; ++i) tab.p[i] = tab.p[0];
    tab.total_blocks = blocks;
    hipLaunchKernelGGL(wgrad_grouped_kernel, dim3(blocks), dim3(THREADS), 0, air_stream(stream), tab);
    AIR_CHECK_LAUNCH();
    return 0;
}
