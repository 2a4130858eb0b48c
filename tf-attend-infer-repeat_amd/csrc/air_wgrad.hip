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


// sum of squares of everything this workgroup stored (tf.global_norm terms): one partial per
// workgroup, reduced again in fixed order by the Adam kernel; workgroup 0 also counts the step
// (apply_gradients(global_step=...), air_model.py:692-694)
__device__ __forceinline__ void publish_sq(float sq, float* sq_partials, int32_t* istate) {
    __shared__ float sq_red[4];
    sq = air_block_sum_256(sq, sq_red);
    if (threadIdx.x == 0) {
        sq_partials[blockIdx.x] = sq;
        if (blockIdx.x == 0 && istate) istate[AIR_IST_GLOBAL_STEP] += 1;
    }
}

__global__ __launch_bounds__(THREADS) void wgrad_grouped_kernel(Table tab, float* __restrict__ sq_partials, int32_t* __restrict__ istate)
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

    // copy the problem descriptor out of the kernel-argument table once
    const float* __restrict__ Ap = pr.A;
    const float* __restrict__ Yp = pr.dY;
    const int M = pr.M, N = pr.N, K = pr.K, lda = pr.lda, ldb = pr.ldb, ldc = pr.ldc;
    float* dW = pr.dW;
    float* db = pr.db;
    const int head_pack = pr.head_pack;

    // staging map: 32 rows x 16 float4 columns = 512 float4 per operand chunk, 2 per thread
    const int srow = tid >> 4, scol = (tid & 15) * 4;             // rows srow and srow + 16
    constexpr int NCH = 8;                                        // chunks in flight: K <= 256 in one round trip
    float4 ra[NCH][2], rb[NCH][2];
    auto fetch = [&](int set, int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + srow + 16 * h;
            ra[set][h] = load4(Ap, lda, k, m0 + scol, K, M, vecA);
            rb[set][h] = load4(Yp, ldb, k, n0 + scol, K, N, vecB);
        }
    };
    auto stage = [&](int set, int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<float4*>(&As[buf][(srow + 16 * h) * LS + scol]) = ra[set][h];
            *reinterpret_cast<float4*>(&Bs[buf][(srow + 16 * h) * LS + scol]) = rb[set][h];
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float colsum = 0.0f;        // wave 0 of block-row 0: column n0 + lane of dY (or of A for the head units)
    const bool do_bias = (db != nullptr) && (wave == 0) && (head_pack ? (n0 == 0) : (m0 == 0));

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
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (ks0 + c * KC < K) fetch(c, ks0 + c * KC);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (ks0 + c * KC < K) {
                stage(c, c & 1);
                __syncthreads();
                compute(c & 1);
            }
    }

    // epilogue: C/D map row = (lane>>4)*4 + q, col = lane&15.  The tile goes through LDS so that
    // every store instruction writes whole 256-byte rows (full cache lines).
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
    float sq = 0.0f;
    if (!head_pack) {
        const bool vecC = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(dW) & 15) == 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (tid >> 4) + 16 * r, col = (tid & 15) * 4;
            const int m = m0 + row, n = n0 + col;
            if (m >= M) continue;
            const float4 v = *reinterpret_cast<const float4*>(&Ct[row * LS + col]);
            float* dst = dW + (size_t)m * ldc + n;
            if (vecC && n + 3 < N) { *reinterpret_cast<float4*>(dst) = v; sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); }
            else {
                if (n < N) { dst[0] = v.x; sq += v.x * v.x; }
                if (n + 1 < N) { dst[1] = v.y; sq += v.y * v.y; }
                if (n + 2 < N) { dst[2] = v.z; sq += v.z * v.z; }
                if (n + 3 < N) { dst[3] = v.w; sq += v.w * v.w; }
            }
        }
        if (do_bias && n0 + lane < N) { db[n0 + lane] = colsum; sq += colsum * colsum; }
    } else {
        // head output units (air_model.py:294-316, 376): A = d_out7 [K,8], dY = hid [K,HT];
        // unit o only owns the hidden segment of its head: dW = wout[o][n - off], db = bout[o] = sum_k d_out7[k][o]
        const int wid[5] = {pr.Hs, pr.Hs, pr.Hh, pr.Hh, pr.Hz};
        const int head[7] = {0, 1, 2, 2, 3, 3, 4};
        for (int it = tid; it < 7 * BT; it += THREADS) {
            const int o = it / BT, col = it % BT, n = n0 + col;
            int off = 0;
            for (int h = 0; h < head[o]; ++h) off += wid[h];
            if (m0 == 0 && n < N && n >= off && n < off + wid[head[o]]) {
                const float v = Ct[o * LS + col];
                dW[(size_t)o * ldc + (n - off)] = v;
                sq += v * v;
            }
        }
        if (do_bias && lane < 7) { db[lane] = colsum; sq += colsum * colsum; }
    }
    if (sq_partials) publish_sq(sq, sq_partials, istate);
}


// ---------------------------------------------------------------------------
// bf16-operand variant (precision 1): operands are rounded to bf16 (RNE,
// v_cvt_pk_bf16_f32) on their way into LDS, products run on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  K sits on the slow (row)
// axis of both operands, the MFMA wants 8 consecutive k per lane: a thread
// loads the same 4 columns of 8 consecutive rows (8 x 16-B loads), packs one
// 16-B k-run per column and stores it into a [column][k] image whose 16-B slots
// are XOR-swizzled by the column index -- conflict-free for the 8-lane
// ds_write_b128 groups and for the 16-lane ds_read_b128 fragment groups.  All
// loads of up to 192 rows are in flight at once, one barrier before the MFMAs
// (no per-chunk barriers); the bias column sums are taken from the fp32
// registers before rounding.
// ---------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int KB = 64;          // rows per LDS image
constexpr int NIMG = 3;         // images resident per round: K <= 192 needs one round

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void wgrad_grouped_bf16_kernel(Table tab, float* __restrict__ sq_partials, int32_t* __restrict__ istate)
{
    // [operand][image][column 0..63][k 0..63] bf16 = 2 x 3 x 8 KB; reused as the fp32 output tile
    __shared__ __attribute__((aligned(16))) unsigned short Img[2 * NIMG * BT * KB];

    int pi = 0;
    while (pi + 1 < tab.count && (int)blockIdx.x >= tab.p[pi + 1].first_block) ++pi;
    const Prob& pr = tab.p[pi];
    const int local = blockIdx.x - pr.first_block;
    const int m0 = (local / pr.tiles_n) * BT, n0 = (local % pr.tiles_n) * BT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const float* __restrict__ Ap = pr.A;
    const float* __restrict__ Yp = pr.dY;
    const int M = pr.M, N = pr.N, K = pr.K, lda = pr.lda, ldb = pr.ldb, ldc = pr.ldc;
    float* dW = pr.dW;
    float* db = pr.db;
    const int head_pack = pr.head_pack;

    // staging role: threads 0..127 own operand A, 128..255 own dY; g = k-run (8 rows), q = column quad
    const int op = tid >> 7, g = tid & 7, q = (tid & 127) >> 3;
    const float* src = op ? Yp : Ap;
    const int ld = op ? ldb : lda, cols = op ? N : M, c0 = (op ? n0 : m0) + 4 * q;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    unsigned short* img = Img + (size_t)op * NIMG * BT * KB;
    const bool full = vec && ((op ? n0 : m0) + BT <= cols);      // wave-uniform (op is)
    const bool bias_block = (db != nullptr) && (head_pack ? (n0 == 0) : (m0 == 0));
    const bool bias_thread = bias_block && (op == (head_pack ? 0 : 1));
    float csum[4] = {0.f, 0.f, 0.f, 0.f};

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kr = 0; kr < K; kr += NIMG * KB) {
        if (kr > 0) __syncthreads();                     // every wave is done reading the previous images
        float4 v[NIMG][8];
        if (full) {
            // interior tile: uniform base + 32-bit byte offsets, rows past K read as zero
            const char* base = reinterpret_cast<const char*>(src);
            const unsigned off0 = ((unsigned)(kr + g * 8) * (unsigned)ld + (unsigned)c0) * 4u;
#pragma unroll
            for (int c = 0; c < NIMG; ++c)
                if (kr + c * KB < K) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const bool ok = kr + c * KB + g * 8 + r < K;
                        const unsigned off = off0 + (unsigned)(c * KB + r) * ((unsigned)ld * 4u);
                        const float4 t = *reinterpret_cast<const float4*>(base + (ok ? off : 0u));
                        v[c][r] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
        } else {
#pragma unroll
            for (int c = 0; c < NIMG; ++c)
                if (kr + c * KB < K) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[c][r] = load4(src, ld, kr + c * KB + g * 8 + r, c0, K, cols, vec);
                }
        }
#pragma unroll
        for (int c = 0; c < NIMG; ++c)
            if (kr + c * KB < K) {
                if (bias_thread) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) { csum[0] += v[c][r].x; csum[1] += v[c][r].y; csum[2] += v[c][r].z; csum[3] += v[c][r].w; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = 4 * q + j;
                    auto e = [&](int r) { const float4& t = v[c][r]; return j == 0 ? t.x : j == 1 ? t.y : j == 2 ? t.z : t.w; };
                    uint4 w;
                    w.x = pack_bf16(e(0), e(1)); w.y = pack_bf16(e(2), e(3));
                    w.z = pack_bf16(e(4), e(5)); w.w = pack_bf16(e(6), e(7));
                    *reinterpret_cast<uint4*>(&img[(size_t)c * BT * KB + col * KB + ((g ^ (col & 7)) << 3)]) = w;
                }
            }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NIMG; ++c)
            if (kr + c * KB < K) {
                const unsigned short* ai = Img + (size_t)c * BT * KB;
                const unsigned short* bi = Img + (size_t)(NIMG + c) * BT * KB;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int slot = ks * 4 + (lane >> 4);
                    bf16x8 av[2], bv[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int col = wm + i * 16 + (lane & 15);
                        av[i] = *reinterpret_cast<const bf16x8*>(&ai[col * KB + ((slot ^ (col & 7)) << 3)]);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int col = wn + j * 16 + (lane & 15);
                        bv[j] = *reinterpret_cast<const bf16x8*>(&bi[col * KB + ((slot ^ (col & 7)) << 3)]);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
            }
    }

    float sq = 0.0f;
    // bias: reduce the 8 k-runs (lanes g = 0..7 are contiguous) of each column quad
    if (bias_block) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = csum[j];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            csum[j] = s;
        }
        if (bias_thread && g == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = 4 * q + j;
                if (!head_pack) { if (n0 + col < N) { db[n0 + col] = csum[j]; sq += csum[j] * csum[j]; } }
                else if (col < 7) { db[col] = csum[j]; sq += csum[j] * csum[j]; }
            }
        }
    }

    // epilogue through LDS: whole 256-byte rows per store instruction
    __syncthreads();
    float* Ct = reinterpret_cast<float*>(Img);           // 64 x LS floats = 20 KB <= 48 KB
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                Ct[(wm + i * 16 + (lane >> 4) * 4 + qq) * LS + wn + j * 16 + (lane & 15)] = acc[i][j][qq];
    __syncthreads();
    if (!head_pack) {
        const bool vecC = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(dW) & 15) == 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (tid >> 4) + 16 * r, col = (tid & 15) * 4;
            const int m = m0 + row, n = n0 + col;
            if (m >= M) continue;
            const float4 t = *reinterpret_cast<const float4*>(&Ct[row * LS + col]);
            float* dst = dW + (size_t)m * ldc + n;
            if (vecC && n + 3 < N) { *reinterpret_cast<float4*>(dst) = t; sq += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w); }
            else {
                if (n < N) { dst[0] = t.x; sq += t.x * t.x; }
                if (n + 1 < N) { dst[1] = t.y; sq += t.y * t.y; }
                if (n + 2 < N) { dst[2] = t.z; sq += t.z * t.z; }
                if (n + 3 < N) { dst[3] = t.w; sq += t.w * t.w; }
            }
        }
    } else {
        const int wid[5] = {pr.Hs, pr.Hs, pr.Hh, pr.Hh, pr.Hz};
        const int head[7] = {0, 1, 2, 2, 3, 3, 4};
        for (int it = tid; it < 7 * BT; it += THREADS) {
            const int o = it / BT, col = it % BT, n = n0 + col;
            int off = 0;
            for (int h = 0; h < head[o]; ++h) off += wid[h];
            if (m0 == 0 && n < N && n >= off && n < off + wid[head[o]]) {
                const float t = Ct[o * LS + col];
                dW[(size_t)o * ldc + (n - off)] = t;
                sq += t * t;
            }
        }
    }
    if (sq_partials) publish_sq(sq, sq_partials, istate);
}

}  // namespace

static int fill_table(const air_wgrad_t* probs, int count, Table& tab) {
    if (!probs || count <= 0) return AIR_EINVAL;
    if (count > MAXP) return AIR_ELIMIT;
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
    return 0;
}

extern "C" int air_wgrad_num_blocks(const air_wgrad_t* probs, int count) {
    Table tab;
    const int rc = fill_table(probs, count, tab);
    return rc ? rc : tab.total_blocks;
}

extern "C" int air_wgrad_grouped(const air_wgrad_t* probs, int count, int precision,
                                 float* sq_partials, int32_t* istate, void* stream) {
    Table tab;
    const int rc = fill_table(probs, count, tab);
    if (rc) return rc;
    if (precision != 0 && precision != 1) return AIR_EINVAL;
    if (precision == 1) hipLaunchKernelGGL(wgrad_grouped_bf16_kernel, dim3(tab.total_blocks), dim3(THREADS), 0, air_stream(stream), tab, sq_partials, istate);
    else hipLaunchKernelGGL(wgrad_grouped_kernel, dim3(tab.total_blocks), dim3(THREADS), 0, air_stream(stream), tab, sq_partials, istate);
    AIR_CHECK_LAUNCH();
    return 0;
}
