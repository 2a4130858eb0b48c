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
#include "air_wgrad_tile.h"
#include <cstring>
#include <cstdlib>

using namespace airw;

namespace {

// fp32 tile: the 64 x 64 block (m0, n0) of pr.A^T . pr.dY, left as 64 rows of LS floats at &As[0][0]
// (barrier-synchronised).  Returns this thread's share of the squared bias gradient it stored.
template <int NCH>      // chunks of 32 rows in flight per round trip (8: K <= 256 in one)
__device__ __forceinline__ float tile_f32(const Prob& pr, int m0, int n0, float (*As)[KC * LS], float (*Bs)[KC * LS])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;        // this wave's quadrant
    const bool vecA = ((pr.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(pr.A) & 15) == 0);
    const bool vecB = ((pr.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(pr.dY) & 15) == 0);

    // copy the problem descriptor out of the kernel-argument table once
    const float* __restrict__ Ap = pr.A;
    const float* __restrict__ Yp = pr.dY;
    const int M = pr.M, N = pr.N, K = pr.K, lda = pr.lda, ldb = pr.ldb;
    float* db = pr.db;
    const int head_pack = pr.head_pack;

    // staging map: 32 rows x 16 float4 columns = 512 float4 per operand chunk, 2 per thread
    const int srow = tid >> 4, scol = (tid & 15) * 4;             // rows srow and srow + 16
    float4 ra[NCH][2], rb[NCH][2];
    auto fetch = [&](int set, int k0, auto vec_t) __attribute__((always_inline)) {
        constexpr bool V = decltype(vec_t)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + srow + 16 * h;
            ra[set][h] = fetch4<V>(Ap, lda, k, m0 + scol, K, M);
            rb[set][h] = fetch4<V>(Yp, ldb, k, n0 + scol, K, N);
        }
    };
    // interior tile, whole chunks: nothing to clamp or mask
    const bool plain = vecA && vecB && m0 + BT <= M && n0 + BT <= N && (K % KC) == 0;
    auto fetch_plain = [&](int set, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned k = (unsigned)(k0 + srow + 16 * h);
            ra[set][h] = *reinterpret_cast<const float4*>(Ap + (k * (unsigned)lda + (unsigned)(m0 + scol)));
            rb[set][h] = *reinterpret_cast<const float4*>(Yp + (k * (unsigned)ldb + (unsigned)(n0 + scol)));
        }
    };
    auto stage = [&](int set, int buf, int k0, auto plain_t) __attribute__((always_inline)) {
        constexpr bool PL = decltype(plain_t)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + srow + 16 * h;
            *reinterpret_cast<float4*>(&As[buf][(srow + 16 * h) * LS + scol]) = PL ? ra[set][h] : mask4(ra[set][h], k, m0 + scol, K, M);
            *reinterpret_cast<float4*>(&Bs[buf][(srow + 16 * h) * LS + scol]) = PL ? rb[set][h] : mask4(rb[set][h], k, n0 + scol, K, N);
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float colsum = 0.0f;        // wave 0 of block-row 0: column n0 + lane of dY (or of A for the head units)
    const bool do_bias = (db != nullptr) && (wave == 0) && (head_pack ? (n0 == 0) : owns_bias(pr, m0, n0));

    auto compute = [&](int buf) {
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
            const float* cs = head_pack ? as : bs;                 // zero rows beyond K
#pragma unroll 8
            for (int k = 0; k < KC; ++k) colsum += cs[k * LS + lane];
        }
    };
    // ALL loads of up to NCH chunks are issued before the first is consumed (one memory round
    // trip), then one barrier per chunk: stage(c+1) only overwrites the buffer every wave finished
    // reading before it passed the barrier of iteration c.
    for (int ks0 = 0; ks0 < K; ks0 += NCH * KC) {
        if (ks0 > 0) __syncthreads();
        if (plain) {                            // ONE uniform branch around the whole batch of loads
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (ks0 + c * KC < K) fetch_plain(c, ks0 + c * KC);
        } else if (vecA && vecB) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (ks0 + c * KC < K) fetch(c, ks0 + c * KC, std::true_type{});
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (ks0 + c * KC < K) fetch(c, ks0 + c * KC, std::false_type{});
        }
        if (plain) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (ks0 + c * KC < K) {
                    stage(c, c & 1, ks0 + c * KC, std::true_type{});
                    __syncthreads();
                    compute(c & 1);
                }
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (ks0 + c * KC < K) {
                    stage(c, c & 1, ks0 + c * KC, std::false_type{});
                    __syncthreads();
                    compute(c & 1);
                }
        }
    }

    // C/D map row = (lane>>4)*4 + q, col = lane&15.  The tile goes through LDS so that every store
    // instruction of the epilogue touches whole 256-byte rows (full cache lines).
    __syncthreads();
    float* Ct = &As[0][0];                      // 64 x LS floats = 20 KB: spans As[0..1]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Ct[(wm + i * 16 + (lane >> 4) * 4 + q) * LS + wn + j * 16 + (lane & 15)] = acc[i][j][q];
    __syncthreads();
    float bias_sq = 0.0f;
    if (do_bias) {
        if (!head_pack) { if (n0 + lane < N) { db[n0 + lane] = colsum; bias_sq = colsum * colsum; } }
        else if (lane < 7) { db[lane] = colsum; bias_sq = colsum * colsum; }
    }
    return bias_sq;
}

__global__ __launch_bounds__(THREADS) void wgrad_grouped_kernel(Table tab, float* __restrict__ sq_partials, int32_t* __restrict__ istate)
{
    __shared__ __attribute__((aligned(16))) float As[2][KC * LS];
    __shared__ __attribute__((aligned(16))) float Bs[2][KC * LS];
    const int block = (int)blockIdx.x;
    __shared__ float sq_red[4 + 64];                            // (+ 64: a bias workgroup parks half of its column sums)
    if (block < tab.nbias) { run_bias_block(tab, block, sq_partials, sq_red); return; }
    int m0, n0;
    const Prob& pr = find_tile(tab, block, m0, n0);
    const float bias_sq = tile_f32<6>(pr, m0, n0, As, Bs);      // K = 192 (3 steps x 64 images) in one round trip
    const float sq = store_tile(pr, m0, n0, &As[0][0], 0.0f) + bias_sq;
    if (sq_partials) publish_sq(sq, sq_partials, istate, pr.first_part + (block - pr.first_block), sq_red);
}


__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void wgrad_grouped_bf16_kernel(
    Table tab, float* __restrict__ sq_partials, int32_t* __restrict__ istate)
{
    // [operand][image][column 0..63][k 0..63] bf16 = 2 x 3 x 8 KB; reused as the fp32 output tile
    // dynamic: 48 KB of operand images (a strip workgroup uses 32), then STRIP_TAIL bytes: 4 floats for the partial's
    // reduction + the parked bias squares of a strip
    extern __shared__ __attribute__((aligned(16))) unsigned short Img[];
    float* sq_red = reinterpret_cast<float*>(Img + 2 * NIMG_W * BT * KB);
    run_tile_bf16(tab, (int)blockIdx.x, Img, sq_partials, istate, sq_red);
}

}  // namespace

AIR_STAMPS_READER(air_debug_stamps_wgrad)
#ifdef AIR_STAMPS
extern "C" int air_debug_stamps_wgrad_wg(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(airw::air_wgrad_wg_stamps), sizeof(unsigned long long) * (n < 2048 * 4 ? n : 2048 * 4));
}
#endif

// Tiles from which a problem counts as BIG (shape alone): it runs in strips when it has twins, and the owners of its bias
// columns are spread over block-rows in every precision.
constexpr long BIG_TILES = 512;
// A problem runs in strips (run_strip_bf16) when its output dwarfs its operands: 4 column tiles per workgroup from 2048
// tiles (dWx at 128 x 128: 4096 tiles over K = 256), 2 from 512 (dWx at 50 x 50: 640 light K = 64 tiles that, one per
// workgroup, pushed the launch past the 768 resident workgroups into a second round).
static int strip_of(const air_wgrad_t& g, bool allow) {
    if (!allow || !g.A16 || !g.dY16 || !g.dW || g.head_pack) return 0;
    if (g.K != 64 && g.K != 128 && g.K != 192 && g.K != 256) return 0;
    if ((g.lda & 3) != 0 || (g.M & 3) != 0 || ((uintptr_t)g.A16 & 7) != 0) return 0;
    if ((g.ldb & 7) != 0 || ((uintptr_t)g.dY16 & 15) != 0) return 0;
    const long tiles = (long)((g.M + BT - 1) / BT) * ((g.N + BT - 1) / BT);
    if (tiles < BIG_TILES) return 0;
    int w = tiles >= 2048 ? 4 : 2;
    if (w > STRIP_MAXG) w = STRIP_MAXG;
    while (w > 1 && (g.N % (BT * w)) != 0) w >>= 1;
    return w > 1 ? w : 0;
}

static int fill_table(const air_wgrad_t* probs, int count, Table& tab, bool allow_null_dw, bool strips = false) {
    if (!probs || count <= 0) return AIR_EINVAL;
    if (count > MAXP) return AIR_ELIMIT;
    tab.count = count;
    // bias columns of the long contractions (K >= 384, shape alone: every precision and operand path agrees) are summed by
    // workgroups of their own, the launch's first (run_bias_wg); their partials follow the tiles'
    int nbias = 0;
    for (int i = 0; i < count; ++i) {
        const air_wgrad_t& g = probs[i];
        if (g.db && !g.head_pack && g.K >= 384 && g.N > 0) nbias += (g.N + BT - 1) / BT;
    }
    tab.nbias = nbias;
    int blocks = nbias, parts = 0, bias_blocks = 0;
    for (int i = 0; i < count; ++i) {
        const air_wgrad_t& g = probs[i];
        if (!g.A || !g.dY || g.M <= 0 || g.N <= 0 || g.K <= 0) return AIR_EINVAL;
        if (!g.dW && (g.head_pack || !allow_null_dw)) return AIR_EINVAL;     // norm-only problems: plain layout, and only with sq_partials
        Prob& p = tab.p[i];
        p.A = g.A; p.dY = g.dY; p.dW = g.dW; p.db = g.db;
        p.A16 = g.A16; p.dY16 = g.dY16;
        p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc;
        p.head_pack = g.head_pack; p.Hs = g.Hs; p.Hh = g.Hh; p.Hz = g.Hz;
        p.tiles_n = (g.N + BT - 1) / BT;
        p.first_block = blocks;
        p.first_part = parts;
        p.bias_first = -1; p.bias_part = -1;
        if (g.db && !g.head_pack && g.K >= 384) { p.bias_first = bias_blocks; p.bias_part = bias_blocks; bias_blocks += p.tiles_n; }
        p.strip = strip_of(g, strips);
        const int tiles_m = (g.M + BT - 1) / BT;
        p.bias_mod = (!g.head_pack && (long)tiles_m * p.tiles_n >= BIG_TILES) ? (tiles_m < 16 ? tiles_m : 16) : 0;
        tab.first[i] = blocks;
        const int tiles = p.tiles_n * ((g.M + BT - 1) / BT);
        parts += tiles;
        blocks += p.strip ? tiles / p.strip : tiles;
    }
    for (int i = count; i < MAXP; ++i) { tab.p[i] = tab.p[0]; tab.first[i] = 0x7fffffff; }
    for (int i = 0; i < count; ++i)
        if (tab.p[i].bias_part >= 0) tab.p[i].bias_part += parts;       // the bias partials follow the tiles'
    tab.total_blocks = parts + nbias;
    tab.launch_blocks = blocks;
    return 0;
}

extern "C" int air_wgrad_num_blocks(const air_wgrad_t* probs, int count) {
    Table tab;
    const int rc = fill_table(probs, count, tab, true);
    return rc ? rc : tab.total_blocks;
}

extern "C" int air_wgrad_num_workgroups(const air_wgrad_t* probs, int count, int precision) {
    if (precision != 0 && precision != 1) return AIR_EINVAL;
    Table tab;
    const int rc = fill_table(probs, count, tab, true, precision == 1);
    return rc ? rc : tab.launch_blocks;
}

extern "C" int air_wgrad_grouped(const air_wgrad_t* probs, int count, int precision,
                                 float* sq_partials, int32_t* istate, void* stream) {
    if (precision != 0 && precision != 1) return AIR_EINVAL;
    Table tab;
    const int rc = fill_table(probs, count, tab, sq_partials != nullptr, precision == 1);
    if (rc) return rc;
    if (precision == 1) {
        static_assert(STRIP_LDS <= 2 * NIMG_W * BT * KB * 2, "a strip image must fit the one-tile workgroups' LDS");
        const size_t lds = sizeof(unsigned short) * 2 * NIMG_W * BT * KB + STRIP_TAIL;
        const int rg = air_grant_lds(reinterpret_cast<const void*>(wgrad_grouped_bf16_kernel), lds);
        if (rg) return rg;
        hipLaunchKernelGGL(wgrad_grouped_bf16_kernel, dim3(tab.launch_blocks), dim3(THREADS), lds, air_stream(stream), tab, sq_partials, istate);
    }
    else hipLaunchKernelGGL(wgrad_grouped_kernel, dim3(tab.launch_blocks), dim3(THREADS), 0, air_stream(stream), tab, sq_partials, istate);
    AIR_CHECK_LAUNCH();
    return 0;
}

