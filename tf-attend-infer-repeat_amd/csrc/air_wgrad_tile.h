// Device code of the weight-gradient tiles (dW = A^T . dY, one 64 x 64 output tile per workgroup) of the grouped
// weight-gradient launch (air_wgrad.hip).  A header of its own since round 4, when the tiles could also RIDE as trailing
// workgroups of the narrow GEMM launches at the end of the backward chain (dh_heads, the BPTT steps) -- built, bit-identical,
// and measured slower (a launch lasts as long as its slowest workgroup, and a K = 192 tile outlasts the carrying products:
// 0.1912 vs 0.1762 ms per step; the rider code alone cost the carriers 1.4 us through its registers): removed again,
// DESIGN.md section 8.
#pragma once
#include "air_common.h"
#include <type_traits>

namespace airw {


typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 256;
constexpr int BT = 64;          // output tile (both dims)
constexpr int KC = 32;          // rows per staged chunk
constexpr int LS = BT + 16;     // LDS row stride: = 16 (mod 32) dwords -> conflict-free fragment reads, 16-B aligned
constexpr int MAXP = 16;

struct Prob {
    const float* A; const float* dY; float* dW; float* db;
    const unsigned short* A16; const unsigned short* dY16;      // bf16 twins of A / dY (nullable): see tile_bf16_tw
    int M, N, K, lda, ldb, ldc;
    int head_pack, Hs, Hh, Hz;
    int tiles_n, first_block;   // first_block: first WORKGROUP of the problem in the grouped launch
    int first_part;             // first global-norm partial (= tile number: one partial per 64 x 64 tile, whoever computes it)
    int strip;                  // > 0: a workgroup owns `strip` consecutive column tiles of one block-row (run_strip_bf16)
    int bias_mod;               // which block-row sums db of column tile nt: 0 -> block-row 0; else block-row nt % bias_mod
    int bias_first, bias_part;  // >= 0: db is summed by workgroups OF ITS OWN (run_bias_wg), one per column tile -- their first
                                // workgroup in the launch and their first global-norm partial; -1: by the tiles (owns_bias)
};
// db[n0 .. n0 + 63] (and its share of the global norm) belongs to ONE tile of the column.  Block-row 0 for ordinary problems;
// for the big ones (>= 2048 tiles, by shape alone -- so every precision / operand path of a problem agrees and the partials
// stay bit-identical between them) the owners are spread over the first block-rows: sixteen column sums in block-row 0
// were the long pole of a strip launch (a strip workgroup of block-row 0 summed all of its columns: +12 us at 4 tiles).
__device__ __forceinline__ bool owns_bias(const Prob& pr, int m0, int n0) {
    if (pr.bias_first >= 0) return false;
    const int nt = n0 / BT, mb = m0 / BT;
    return pr.bias_mod > 0 ? (mb == nt % pr.bias_mod) : (mb == 0);
}
// first[i] = first workgroup of problem i (INT_MAX past `count`): kept apart from the descriptors so that
// ONE wide scalar load fetches all of them and the owner is found without a chain of dependent loads
// (total_blocks = tiles = global-norm partials; launch_blocks = workgroups of the grouped launch: fewer when a problem
// runs in strips)
// nbias: the launch's first workgroups, which sum bias columns of the long-contraction problems (run_bias_wg)
struct Table { int count; int total_blocks; int launch_blocks; int nbias; int first[MAXP]; Prob p[MAXP]; };

// One float4 of a row-major operand, zero outside [rows x cols], in two branch-free halves: fetch4
// issues the load(s) with out-of-range accesses redirected to element 0, mask4 zeroes what was out of
// range.  Callers issue ALL fetches of a batch before the first mask: a run-time branch around a
// load (or a consumer right behind it) makes the compiler wait on the spot, which serialises the
// 16-24 loads a thread should have in flight (12 us instead of 3 for the ragged-edge tiles).
// VEC: base 16-byte aligned and ld % 4 == 0 -- a quad that starts inside a row then lies inside
// the padded row, so the 16-byte load is issued even when its last columns are past `cols`.
template <bool VEC>
__device__ __forceinline__ float4 fetch4(const float* __restrict__ base, int ld, int row, int col, int rows, int cols) {
    const bool okr = row < rows;
    const unsigned at = (unsigned)row * (unsigned)ld + (unsigned)col;
    if (VEC) return *reinterpret_cast<const float4*>(base + ((okr && col < cols) ? at : 0u));
    float4 v;
    v.x = base[(okr && col < cols) ? at : 0u];
    v.y = base[(okr && col + 1 < cols) ? at + 1u : 0u];
    v.z = base[(okr && col + 2 < cols) ? at + 2u : 0u];
    v.w = base[(okr && col + 3 < cols) ? at + 3u : 0u];
    return v;
}
__device__ __forceinline__ float4 mask4(float4 t, int row, int col, int rows, int cols) {
    const bool okr = row < rows;
    float4 v;
    v.x = (okr && col < cols) ? t.x : 0.f;
    v.y = (okr && col + 1 < cols) ? t.y : 0.f;
    v.z = (okr && col + 2 < cols) ? t.z : 0.f;
    v.w = (okr && col + 3 < cols) ? t.w : 0.f;
    return v;
}


// sum of squares of everything this workgroup stored (tf.global_norm terms): one partial per
// workgroup, reduced again in fixed order by the Adam kernel; workgroup 0 also counts the step
// (apply_gradients(global_step=...), air_model.py:692-694)
// (sq_red: 4 floats of LDS supplied by the kernel -- a carrying GEMM kernel must not own static LDS: it could no longer
// be granted the whole 160 KB as dynamic LDS, hipFuncSetAttribute refuses static + dynamic > 160 KB)
__device__ __forceinline__ void publish_sq(float sq, float* sq_partials, int32_t* istate, int part, float* sq_red) {
    sq = air_block_sum_256(sq, sq_red);
    if (threadIdx.x == 0) {
        sq_partials[part] = sq;
        if (part == 0 && istate) istate[AIR_IST_GLOBAL_STEP] += 1;
    }
}

// Epilogue of the weight-gradient kernels: the tile in Ct goes to pr.dW (whole rows per store
// instruction) and its sum of squares to the caller.  pr.dW == NULL: nothing is stored, only the
// sum of squares is taken (a gradient its caller rebuilds from the factors elsewhere).
__device__ __forceinline__ float store_tile(const Prob& pr, int m0, int n0, const float* Ct, float sq)
{
    const int tid = threadIdx.x;
    const int M = pr.M, N = pr.N, ldc = pr.ldc;
    float* dW = pr.dW;
    if (!pr.head_pack) {
        const bool vecC = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(dW) & 15) == 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (tid >> 4) + 16 * r, col = (tid & 15) * 4;
            const int m = m0 + row, n = n0 + col;
            if (m >= M) continue;
            const float4 t = *reinterpret_cast<const float4*>(&Ct[row * LS + col]);
            float* dst = dW + (size_t)m * ldc + n;
            if (vecC && n + 3 < N) { if (dW) *reinterpret_cast<float4*>(dst) = t; sq += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w); }
            else {
                if (n < N) { if (dW) dst[0] = t.x; sq += t.x * t.x; }
                if (n + 1 < N) { if (dW) dst[1] = t.y; sq += t.y * t.y; }
                if (n + 2 < N) { if (dW) dst[2] = t.z; sq += t.z * t.z; }
                if (n + 3 < N) { if (dW) dst[3] = t.w; sq += t.w * t.w; }
            }
        }
    } else {
        // head output units (air_model.py:294-316, 376): A = d_out7 [K,8], dY = hid [K,HT];
        // unit o only owns the hidden segment of its head: dW = wout[o][n - off], db = bout[o] = sum_k d_out7[k][o]
        // (unit -> head -> segment as arithmetic: indexed tables cost a chain of dependent loads per element)
        const int Hs = pr.Hs, Hh = pr.Hh, Hz = pr.Hz;
        for (int it = tid; it < 7 * BT; it += THREADS) {
            const int o = it / BT, col = it % BT, n = n0 + col;
            const int hd = o < 2 ? o : (o < 4 ? 2 : (o < 6 ? 3 : 4));
            const int off = hd == 0 ? 0 : hd == 1 ? Hs : hd == 2 ? 2 * Hs : hd == 3 ? 2 * Hs + Hh : 2 * Hs + 2 * Hh;
            const int wd = hd < 2 ? Hs : (hd < 4 ? Hh : Hz);
            if (m0 == 0 && n < N && n >= off && n < off + wd) {
                const float t = Ct[o * LS + col];
                dW[(size_t)o * ldc + (n - off)] = t;
                sq += t * t;
            }
        }
    }
    return sq;
}

__device__ __forceinline__ const Prob& find_tile(const Table& tab, int block, int& m0, int& n0) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAXP; ++i) pi += (block >= tab.first[i]) ? 1 : 0;
    const Prob& pr = tab.p[pi];
    const int local = block - pr.first_block;
    if (pr.strip > 0) {
        // strips: the groups of one block-row share its A block.  Workgroups go round-robin over the 8 XCDs, so block-rows
        // are dealt per XCD (every group of a block-row on the SAME XCD, back to back: one L2 fetches the A block once)
        const int groups = pr.tiles_n / pr.strip, tiles_m = (pr.M + BT - 1) / BT;
        int mb, g;
        if ((tiles_m & 7) == 0) { const int x = local & 7, s = local >> 3; mb = (s / groups) * 8 + x; g = s % groups; }
        else { mb = local / groups; g = local % groups; }
        m0 = mb * BT; n0 = g * pr.strip * BT;
        return pr;
    }
    m0 = (local / pr.tiles_n) * BT; n0 = (local % pr.tiles_n) * BT;
    return pr;
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
constexpr int NIMG_W = 3;       // images resident per round in the weight-gradient kernel: K <= 192 needs one round

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// bf16 tile: as tile_f32, operands rounded to bf16; Img = [operand][image][column][k] shorts of LDS,
// the fp32 tile is left at its start (64 rows of LS floats, barrier-synchronised).
template <int NIMG>     // LDS images (64 rows each) resident per round; Img holds max(2 * NIMG * 8 KB, 20 KB)
__device__ __forceinline__ float tile_bf16(const Prob& pr, int m0, int n0, unsigned short* Img)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const float* __restrict__ Ap = pr.A;
    const float* __restrict__ Yp = pr.dY;
    const int M = pr.M, N = pr.N, K = pr.K, lda = pr.lda, ldb = pr.ldb;
    float* db = pr.db;
    const int head_pack = pr.head_pack;

    // staging role: threads 0..127 own operand A, 128..255 own dY; g = k-run (8 rows), q = column quad
    const int op = tid >> 7, g = tid & 7, q = (tid & 127) >> 3;
    const float* src = op ? Yp : Ap;
    const int ld = op ? ldb : lda, cols = op ? N : M, c0 = (op ? n0 : m0) + 4 * q;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    unsigned short* img = Img + (size_t)op * NIMG * BT * KB;
    const bool bias_block = (db != nullptr) && (head_pack ? (n0 == 0) : owns_bias(pr, m0, n0));
    const bool bias_thread = bias_block && (op == (head_pack ? 0 : 1));
    float csum[4] = {0.f, 0.f, 0.f, 0.f};

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    AIR_STAMP(1);
    for (int kr = 0; kr < K; kr += NIMG * KB) {
        if (kr > 0) __syncthreads();                     // every wave is done reading the previous images
        float4 v[NIMG][8];
        // wave-uniform choices, each ONE branch around the whole batch of loads
        const bool inside = (op ? n0 : m0) + BT <= cols;                  // no ragged column edge
        if (vec && inside && (K % KB) == 0) {
            // interior tile, whole images: uniform base + 32-bit byte offsets, nothing to mask
            const char* base = reinterpret_cast<const char*>(src);
            const unsigned step = (unsigned)ld * 4u;
            const unsigned off0 = (unsigned)(kr + g * 8) * step + (unsigned)c0 * 4u;
#pragma unroll
            for (int c = 0; c < NIMG; ++c)
                if (kr + c * KB < K) {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        v[c][r] = *reinterpret_cast<const float4*>(base + (off0 + (unsigned)(c * KB + r) * step));
                }
        } else if (vec) {
#pragma unroll
            for (int c = 0; c < NIMG; ++c)
                if (kr + c * KB < K) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[c][r] = fetch4<true>(src, ld, kr + c * KB + g * 8 + r, c0, K, cols);
                }
        } else {
#pragma unroll
            for (int c = 0; c < NIMG; ++c)
                if (kr + c * KB < K) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[c][r] = fetch4<false>(src, ld, kr + c * KB + g * 8 + r, c0, K, cols);
                }
        }
        AIR_STAMP(2);
        if (!(vec && inside && (K % KB) == 0)) {
#pragma unroll
            for (int c = 0; c < NIMG; ++c)
                if (kr + c * KB < K) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[c][r] = mask4(v[c][r], kr + c * KB + g * 8 + r, c0, K, cols);
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
        AIR_STAMP(3);
        __syncthreads();
        AIR_STAMP(4);
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
    AIR_STAMP(5);
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
    return sq;
}

// bf16-TWIN tile: both operands are read from the bf16 twins their producers wrote (air_wgrad_t.A16 / dY16).
// K is the slow axis of both, i.e. both are "n-contiguous" for the MFMA: a 64-row image of an operand is
// copied as it lies into a [k][64 columns] LDS image (16- or 8-byte pieces, lane-linear rows of 128 bytes,
// no conversion, no register transpose) and the MFMA fragments -- 8 consecutive k per lane -- come out of
// gfx950's transpose read ds_read_b64_tr_b16 (tools/exp/tr_read.hip; the same scheme as the forward GEMM's
// row-major weights, air_gemm_bf16.hip).  Same bf16 values, same k order per wave as tile_bf16: the tile is
// bit-identical.  The bias gradient still comes from the fp32 dY (column sums before rounding, in tile_bf16's
// order) -- only the tiles that own a column's bias (owns_bias) pay those loads.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bool twin_ok(const Prob& pr) {
    auto a8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
    // (M % 4 != 0 is fine when the A rows are padded to a multiple of 4: a piece that straddles M stays inside its row, and
    // the output rows it feeds are beyond M -- never stored)
    return pr.A16 && pr.dY16 && !pr.head_pack && a8(pr.A16) && a8(pr.dY16) && (pr.lda & 3) == 0 && (pr.ldb & 3) == 0 &&
           ((pr.M & 3) == 0 || pr.lda >= ((pr.M + 3) & ~3)) && (pr.N & 3) == 0;
}

// Bias gradient of a twin tile in block-row 0: fp32 column sums of dY[:, n0 .. n0 + 63] in tile_bf16's order -- per k-run g the
// rows g*8 + r of image c, images in order; the xor-tree over g follows.  Run after the MFMAs (the operand registers are
// free), the next image's 8 loads in flight while this one is summed.  (Round 4 tried three images in flight, all fetches
// unconditional: 36.8 -> 31.9 us for a K = 1280 tile that owns a bias, but 9.4 -> 9.8 us at K = 192 and +1.5 us on the
// 50 x 50 step -- not kept.  A K = 1280 tile is 15.7 us without its bias: the fp32 column sums of 1280 rows by 128 threads
// remain the long pole of the 128 x 128 launch.)  Threads 128..255 hold dY's staging
// role of tile_bf16 (g = k-run of 8 rows, q = column quad); returns this thread's share of the squared bias gradient.
__device__ __forceinline__ float bias_tw(const Prob& pr, int n0)
{
    const int tid = threadIdx.x;
    const int N = pr.N, K = pr.K, ldb = pr.ldb;
    float* db = pr.db;
    const int bg = tid & 7, bq = (tid & 127) >> 3;
    const bool bias_thread = tid >= 128;
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias_thread) {
        const bool vec = ((ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(pr.dY) & 15) == 0);
        const int nimg = (K + KB - 1) / KB;
        float4 cur[8], nxt[8];
        auto fetch_img = [&](float4 (&v)[8], int c) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                v[r] = vec ? fetch4<true>(pr.dY, ldb, c * KB + bg * 8 + r, n0 + 4 * bq, K, N)
                           : fetch4<false>(pr.dY, ldb, c * KB + bg * 8 + r, n0 + 4 * bq, K, N);
        };
        fetch_img(cur, 0);
        for (int c = 0; c < nimg; ++c) {
            if (c + 1 < nimg) fetch_img(nxt, c + 1);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float4 t = mask4(cur[r], c * KB + bg * 8 + r, n0 + 4 * bq, K, N);
                csum[0] += t.x; csum[1] += t.y; csum[2] += t.z; csum[3] += t.w;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) cur[r] = nxt[r];
        }
    }
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s = csum[j];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
        csum[j] = s;
    }
    if (bias_thread && bg == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = 4 * bq + j;
            if (n0 + col < N) { db[n0 + col] = csum[j]; sq += csum[j] * csum[j]; }
        }
    }
    return sq;
}

template <int NIMG>
__device__ __forceinline__ float tile_bf16_tw(const Prob& pr, int m0, int n0, unsigned short* Img)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int M = pr.M, N = pr.N, K = pr.K, lda = pr.lda, ldb = pr.ldb;
    float* db = pr.db;
    const char* Ab = reinterpret_cast<const char*>(pr.A16);
    const char* Yb = reinterpret_cast<const char*>(pr.dY16);
    unsigned short* ImgA = Img;                              // [NIMG][64 k][64 m]
    unsigned short* ImgB = Img + NIMG * KB * BT;             // [NIMG][64 k][64 n]
    // 16-byte pieces when rows start 16-byte aligned (ld % 8 == 0), else 8-byte pieces (ld % 4 == 0)
    const bool a16 = (lda & 7) == 0 && (reinterpret_cast<uintptr_t>(pr.A16) & 15) == 0 && (M & 7) == 0;
    const bool b16 = (ldb & 7) == 0 && (reinterpret_cast<uintptr_t>(pr.dY16) & 15) == 0 && (N & 7) == 0;
    const bool bias_block = (db != nullptr) && owns_bias(pr, m0, n0);

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int T8 = NIMG * KB * 16 / THREADS;             // 8-byte pieces per thread per operand (two of them = one 16-byte piece)
    uint2 ra[T8], rb[T8];
    // all operand loads of one round, one wave-uniform branch per operand.  Rounds are software-pipelined: the loads of
    // round r+1 are issued as soon as round r's registers are in LDS, i.e. under its barrier and MFMAs (K > 192: the
    // 128x128 configuration contracts over 256 rows)
    auto issue_round = [&](int kr) __attribute__((always_inline)) {
        if (a16) {
#pragma unroll
            for (int i = 0; i < T8 / 2; ++i) {
                const int t = tid + THREADS * i, k = kr + (t >> 3), col = m0 + (t & 7) * 8;
                const bool ok = k < K && col < M;
                const uint4 x = *reinterpret_cast<const uint4*>(Ab + (ok ? ((unsigned)k * (unsigned)lda + (unsigned)col) * 2u : 0u));
                ra[2 * i] = ok ? make_uint2(x.x, x.y) : make_uint2(0u, 0u);
                ra[2 * i + 1] = ok ? make_uint2(x.z, x.w) : make_uint2(0u, 0u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < T8; ++i) {
                const int t = tid + THREADS * i, k = kr + (t >> 4), col = m0 + (t & 15) * 4;
                const bool ok = k < K && col < M;
                const uint2 x = *reinterpret_cast<const uint2*>(Ab + (ok ? ((unsigned)k * (unsigned)lda + (unsigned)col) * 2u : 0u));
                ra[i] = ok ? x : make_uint2(0u, 0u);
            }
        }
        if (b16) {
#pragma unroll
            for (int i = 0; i < T8 / 2; ++i) {
                const int t = tid + THREADS * i, k = kr + (t >> 3), col = n0 + (t & 7) * 8;
                const bool ok = k < K && col < N;
                const uint4 x = *reinterpret_cast<const uint4*>(Yb + (ok ? ((unsigned)k * (unsigned)ldb + (unsigned)col) * 2u : 0u));
                rb[2 * i] = ok ? make_uint2(x.x, x.y) : make_uint2(0u, 0u);
                rb[2 * i + 1] = ok ? make_uint2(x.z, x.w) : make_uint2(0u, 0u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < T8; ++i) {
                const int t = tid + THREADS * i, k = kr + (t >> 4), col = n0 + (t & 15) * 4;
                const bool ok = k < K && col < N;
                const uint2 x = *reinterpret_cast<const uint2*>(Yb + (ok ? ((unsigned)k * (unsigned)ldb + (unsigned)col) * 2u : 0u));
                rb[i] = ok ? x : make_uint2(0u, 0u);
            }
        }
    };
    AIR_STAMP(1);
    issue_round(0);
    AIR_STAMP(2);
    for (int kr = 0; kr < K; kr += NIMG * KB) {
        if (kr > 0) __syncthreads();
        // ---- straight into the [k][64] images (lane-linear rows)
        if (a16) {
#pragma unroll
            for (int i = 0; i < T8 / 2; ++i)
                *reinterpret_cast<uint4*>(&ImgA[(tid + THREADS * i) * 8]) = make_uint4(ra[2 * i].x, ra[2 * i].y, ra[2 * i + 1].x, ra[2 * i + 1].y);
        } else {
#pragma unroll
            for (int i = 0; i < T8; ++i) *reinterpret_cast<uint2*>(&ImgA[(tid + THREADS * i) * 4]) = ra[i];
        }
        if (b16) {
#pragma unroll
            for (int i = 0; i < T8 / 2; ++i)
                *reinterpret_cast<uint4*>(&ImgB[(tid + THREADS * i) * 8]) = make_uint4(rb[2 * i].x, rb[2 * i].y, rb[2 * i + 1].x, rb[2 * i + 1].y);
        } else {
#pragma unroll
            for (int i = 0; i < T8; ++i) *reinterpret_cast<uint2*>(&ImgB[(tid + THREADS * i) * 4]) = rb[i];
        }
        AIR_STAMP(3);
        if (kr + NIMG * KB < K) issue_round(kr + NIMG * KB);
        __syncthreads();
        AIR_STAMP(4);
        // ---- MFMAs: fragments through the transpose read (lane i of a 16-lane group hands in row 8g + i/4 (+4), column quad i%4)
        const int il = lane & 15;
#pragma unroll
        for (int c = 0; c < NIMG; ++c)
            if (kr + c * KB < K) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int krow = c * KB + ks * 32 + (lane >> 4) * 8 + (il >> 2);
                    const unsigned short* pa = &ImgA[krow * BT + wm + (il & 3) * 4];
                    const unsigned short* pb = &ImgB[krow * BT + wn + (il & 3) * 4];
                    bf16x8 av[2], bv[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + i * 16));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + i * 16 + 4 * BT));
                        av[i] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb + j * 16));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb + j * 16 + 4 * BT));
                        bv[j] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
            }
    }
    AIR_STAMP(5);
    const float sq = bias_block ? bias_tw(pr, n0) : 0.0f;
    __syncthreads();
    float* Ct = reinterpret_cast<float*>(Img);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                Ct[(wm + i * 16 + (lane >> 4) * 4 + qq) * LS + wn + j * 16 + (lane & 15)] = acc[i][j][qq];
    __syncthreads();
    return sq;
}


// STRIP workgroup (round 4): the 64 x 64 tiles (m0, n0), (m0, n0 + 64), ... of `pr.strip` consecutive column tiles of one
// block-row, for a problem whose output is far larger than its operands -- dWx = X^T . (sum_t dgates) at 128 x 128 is
// 16 384 x 1 024 = 4 096 tiles over K = 256 rows.  One tile per workgroup made every tile fetch its own 32 KB block of the
// image twin (the 16 tiles of a block-row start together on 8 different XCDs) and pay the whole load - stage - multiply -
// store chain (8.9 us a tile with three resident per CU: 48 us for the launch on its own).  Here the A block is fetched ONCE
// and its MFMA fragments stay in registers; the dY tiles stream through one 32 KB image with the next tile's loads in flight
// under this tile's MFMAs, epilogue and stores (3 us a tile); the output tile passes through the same LDS.  Same bf16 values,
// same k order per accumulator, one global-norm partial per TILE at its old index: bit-identical to the per-tile kernel.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int STRIP_KMAX = 256;
#ifdef AIR_STAMPS
#ifndef STRIP_STAMP_BLOCK
#define STRIP_STAMP_BLOCK 600
#endif
#define STRIP_STAMP(i) do { if (blockIdx.x == STRIP_STAMP_BLOCK && threadIdx.x == 0) air_stamps_dev[i] = wall_clock64(); } while (0)
#else
#define STRIP_STAMP(i) do { } while (0)
#endif
constexpr int STRIP_LDS = STRIP_KMAX * BT * 2;              // bytes: one [K][64] image (32 KB <= the 48 KB of the one-tile workgroups)
constexpr int STRIP_MAXG = 16;                               // column tiles per strip workgroup, at most
constexpr int STRIP_TAIL = (4 + STRIP_MAXG * 16) * 4;       // bytes behind the images: reduction scratch + parked bias squares
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int NPER,     // 16-byte pieces per thread per operand = K / 32 (a template argument: the operand registers must be
                        // statically indexed -- with a run-time count the compiler kept them in scratch memory and waited for
                        // every load on the spot: 12 us to stage the first images)
          bool A8>      // the A block in 8-byte pieces, masked at M: rows of the A twin are only 8-byte aligned and the last
                        // block-row is ragged (the 50 x 50 canvas: M = lda = 2500)
__device__ __forceinline__ void run_strip_bf16(const Prob& pr, int m0, int n0, unsigned short* Img, float* __restrict__ sq_partials,
                                               int32_t* __restrict__ istate, float* sq_red)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int G = pr.strip;
    const unsigned lda = (unsigned)pr.lda, ldb = (unsigned)pr.ldb;
    const char* Ab = reinterpret_cast<const char*>(pr.A16);
    const char* Yb = reinterpret_cast<const char*>(pr.dY16);
    unsigned short* ImgB = Img;                               // [K][64 n] (first the A block, [K][64 m]); then the fp32 output tile
    const bool has_bias = pr.db != nullptr;
    constexpr int PMAX = NPER;

    // Bias gradient of the column tiles this block-row owns (owns_bias: at most one of the strip for the big problems),
    // BEFORE the tile loop: the 16 threads that hold a share of its square park it in LDS.
    float* bias_park = sq_red + 4;                            // [strip][16] floats behind the reduction scratch
    if (has_bias) {
        for (int j = 0; j < G; ++j) {
            if (!owns_bias(pr, m0, n0 + j * BT)) continue;    // block-uniform
            const float b = bias_tw(pr, n0 + j * BT);
            if (tid >= 128 && (tid & 7) == 0) bias_park[j * 16 + ((tid & 127) >> 3)] = b;
        }
        __syncthreads();
    }

    STRIP_STAMP(0);
    // piece t = tid + 256 i: row t >> 3, 8 columns from (t & 7) * 8 -- lane-linear rows of 128 bytes
    const unsigned k0 = (unsigned)(tid >> 3), c8 = (unsigned)(tid & 7) * 8u;
    // (the register images are declared per use, never carried around the tile loop: an array that lives across the
    // loop's back edge stayed in scratch memory, every load waited for on the spot)
    auto issue_b = [&](u32x4 (&rb)[PMAX], int nn) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PMAX; ++i)
            rb[i] = *reinterpret_cast<const u32x4*>(Yb + ((k0 + 32u * i) * ldb + (unsigned)nn + c8) * 2u);
    };
    auto stage_b = [&](const u32x4 (&rb)[PMAX]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PMAX; ++i)
            *reinterpret_cast<u32x4*>(&ImgB[(tid + THREADS * i) * 8]) = rb[i];
    };
    // The strip is walked from a start that depends on the block-row: the 256 rows of a dY tile lie 2 KB (N * 2 bytes)
    // apart, i.e. in ONE L2 channel, and workgroups that start together stay in step -- all of them on the same tile would
    // queue on that channel (measured: 10 us per tile, the launch twice as slow as one tile per workgroup).
    const int mb = m0 / BT;
    const int rot = (mb + (mb >> 3)) % G;
    auto tile_at = [&](int j) { int t = j + rot; if (t >= G) t -= G; return t; };
    // The A block passes through the image ONCE: each wave keeps the fragments of its 32 columns (all K rows: NPER k-steps
    // x 2 column groups x 4 registers = 64 registers at K = 256) for the whole strip, so a tile reads only dY fragments from
    // LDS (the fragment reads, not the MFMAs, bounded a tile: 128 KB per tile per workgroup at 128 bytes per clock) and the
    // workgroup needs ONE 32 KB image -- three workgroups per CU, like the one-tile workgroups of the same launch.
    const int il = lane & 15;
    bf16x8 av[NPER][2];
    {
        u32x4 rb[PMAX];
        if constexpr (!A8) {
            u32x4 ra[PMAX];
#pragma unroll
            for (int i = 0; i < PMAX; ++i)
                ra[i] = *reinterpret_cast<const u32x4*>(Ab + ((k0 + 32u * i) * lda + (unsigned)m0 + c8) * 2u);
            issue_b(rb, n0 + tile_at(0) * BT);
#pragma unroll
            for (int i = 0; i < PMAX; ++i)
                *reinterpret_cast<u32x4*>(&ImgB[(tid + THREADS * i) * 8]) = ra[i];
        } else {
            // piece t = tid + 256 i: row t >> 4, 4 columns from (t & 15) * 4; a piece at or past column M reads element 0
            // and is staged as zeros (M % 4 == 0: pieces are whole)
            u32x2 ra[2 * PMAX];
            const unsigned kq = (unsigned)(tid >> 4), c4 = (unsigned)(tid & 15) * 4u;
            const bool okm = m0 + (int)c4 < pr.M;
#pragma unroll
            for (int i = 0; i < 2 * PMAX; ++i)
                ra[i] = *reinterpret_cast<const u32x2*>(Ab + (okm ? ((kq + 16u * i) * lda + (unsigned)m0 + c4) * 2u : 0u));
            issue_b(rb, n0 + tile_at(0) * BT);
            const u32x2 zero = {0u, 0u};
#pragma unroll
            for (int i = 0; i < 2 * PMAX; ++i)
                *reinterpret_cast<u32x2*>(&ImgB[(tid + THREADS * i) * 4]) = okm ? ra[i] : zero;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < NPER; ++kk) {
            const int krow = kk * 32 + (lane >> 4) * 8 + (il >> 2);
            const unsigned short* pa = &ImgB[krow * BT + wm + (il & 3) * 4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + i * 16));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + i * 16 + 4 * BT));
                av[kk][i] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
        }
        __syncthreads();
        stage_b(rb);
    }
    STRIP_STAMP(1);

    const int part0 = pr.first_part + mb * pr.tiles_n + n0 / BT;
    for (int j = 0; j < G; ++j) {
        const int tj = tile_at(j);
        const int nj = n0 + tj * BT;
        __syncthreads();                                      // the images are staged
        STRIP_STAMP(2 + 6 * j);
        // next dY tile: in flight until this tile is stored.  UNCONDITIONAL (the last round re-reads its own tile): behind
        // a branch the loaded registers merge with their old values, and the compiler waits for the loads right here
        u32x4 rb[PMAX];
        issue_b(rb, n0 + tile_at(j + 1 < G ? j + 1 : j) * BT);
        __builtin_amdgcn_sched_barrier(0);                    // (the scheduler would sink the loads below the MFMAs)
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NPER; ++kk) {                   // k-steps of 32 rows, in tile_bf16_tw's order
            const int krow = kk * 32 + (lane >> 4) * 8 + (il >> 2);
            const unsigned short* pb = &ImgB[krow * BT + wn + (il & 3) * 4];
            bf16x8 bv[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb + jj * 16));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb + jj * 16 + 4 * BT));
                bv[jj] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[kk][i], bv[jj], acc[i][jj], 0, 0, 0);
        }
        STRIP_STAMP(3 + 6 * j);
        const float bias_sq = (has_bias && owns_bias(pr, m0, nj) && tid >= 128 && (tid & 7) == 0) ? bias_park[tj * 16 + ((tid & 127) >> 3)] : 0.0f;
        __syncthreads();                                      // every wave is done reading the dY image
        float* Ct = reinterpret_cast<float*>(ImgB);           // 64 x LS floats = 20 KB <= 32 KB
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    Ct[(wm + i * 16 + (lane >> 4) * 4 + qq) * LS + wn + jj * 16 + (lane & 15)] = acc[i][jj][qq];
        __syncthreads();
        STRIP_STAMP(4 + 6 * j);
        const float sq = store_tile(pr, m0, nj, Ct, bias_sq);
        STRIP_STAMP(5 + 6 * j);
        if (sq_partials) publish_sq(sq, sq_partials, istate, part0 + tj, sq_red);
        STRIP_STAMP(6 + 6 * j);
        if (j + 1 < G) {
            __syncthreads();                                  // the output tile has left LDS
            stage_b(rb);
        }
        STRIP_STAMP(7 + 6 * j);
    }
}

// BIAS workgroup (round 4): db[n0 .. n0 + 63] = column sums of the fp32 dY over all K rows, and its squares as one
// global-norm partial of its own.  For the long contractions (K >= 384: the 128 x 128 configuration's N*B = 1280 rows) the
// column sums were the long pole of the launch when a tile did them on the side -- 128 threads, one 64-row image at a
// time: 21 us on top of the tile's 15.7.  Here they have the launch's FIRST workgroups to themselves (they start at once
// and finish under the tiles), all 256 threads (the two halves take alternate images) and two images in flight per
// thread.  Order of summation (the definition for every operand path and precision of such a problem): thread (h, g, q)
// adds rows 8g .. 8g+7 of its images c = h, h+2, ... in order; xor-tree over g; half 0 + half 1.
__device__ __forceinline__ void run_bias_wg(const Prob& pr, int n0, int part, float* __restrict__ sq_partials, float* sq_red)
{
    const int tid = threadIdx.x;
    const int N = pr.N, K = pr.K;
    const unsigned ldb = (unsigned)pr.ldb;
    const float* __restrict__ Y = pr.dY;
    const int g = tid & 7, q = (tid >> 3) & 15, h = tid >> 7;
    const int col = n0 + 4 * q;
    const bool vec = ((ldb & 3u) == 0u) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
    const int nimg = (K + KB - 1) / KB;
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    auto fetch_img = [&](f32x4 (&v)[8], int c) __attribute__((always_inline)) {
        if (c >= nimg) c = nimg - 1;                          // (past the end: a valid image again, never summed)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = c * KB + g * 8 + r;
            const bool okr = row < K;
            const unsigned at = (unsigned)row * ldb + (unsigned)col;
            if (vec) v[r] = *reinterpret_cast<const f32x4*>(Y + ((okr && col < N) ? at : 0u));
            else {
                v[r].x = Y[(okr && col < N) ? at : 0u];
                v[r].y = Y[(okr && col + 1 < N) ? at + 1u : 0u];
                v[r].z = Y[(okr && col + 2 < N) ? at + 2u : 0u];
                v[r].w = Y[(okr && col + 3 < N) ? at + 3u : 0u];
            }
        }
    };
    auto sum_img = [&](const f32x4 (&v)[8], int c) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const bool okr = c * KB + g * 8 + r < K;
            csum[0] += (okr && col < N) ? v[r].x : 0.f;
            csum[1] += (okr && col + 1 < N) ? v[r].y : 0.f;
            csum[2] += (okr && col + 2 < N) ? v[r].z : 0.f;
            csum[3] += (okr && col + 3 < N) ? v[r].w : 0.f;
        }
    };
    f32x4 v0[8], v1[8];
    fetch_img(v0, h);
    for (int c = h; c < nimg; c += 4) {
        fetch_img(v1, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        sum_img(v0, c);
        fetch_img(v0, c + 4);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < nimg) sum_img(v1, c + 2);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float t = csum[j];
        t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4);
        csum[j] = t;
    }
    float* park = sq_red + 4;                                 // 64 floats: half 1's column sums
    if (h == 1 && g == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) park[4 * q + j] = csum[j];
    }
    __syncthreads();
    float sq = 0.0f;
    if (h == 0 && g == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = csum[j] + park[4 * q + j];
            if (col + j < N) { pr.db[col + j] = t; sq += t * t; }
        }
    }
    if (sq_partials) {
        sq = air_block_sum_256(sq, sq_red);
        if (tid == 0) sq_partials[part] = sq;
    }
}

// the launch's first tab.nbias workgroups: bias columns of the long-contraction problems (either precision)
__device__ __forceinline__ void run_bias_block(const Table& tab, int block, float* __restrict__ sq_partials, float* sq_red) {
    int k = 0;                               // the last problem whose bias workgroups start at or before this one
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
        if (i < tab.count && tab.p[i].bias_first >= 0 && block >= tab.p[i].bias_first) k = i;
    const Prob& pb = tab.p[k];
    const int idx = block - pb.bias_first;
    run_bias_wg(pb, idx * BT, pb.bias_part + idx, sq_partials, sq_red);
}

// One rider / grouped-launch workgroup of the bf16 path: tile `block` of the table; Img = 48 KB of LDS (16-byte aligned;
// STRIP_LDS when the table holds a strip problem), sq_red = 4 more floats.  sq_partials == NULL: no global-norm partial is
// published.
// per-workgroup stamps of the grouped launch (debug builds with -DAIR_STAMPS only; tools/wgrad_wg_stamps.py): [block][4] =
// start, end, hardware id (XCC_ID << 16 | HW_ID), K of its problem
#ifdef AIR_STAMPS
static __device__ unsigned long long air_wgrad_wg_stamps[2048 * 4];
#define WG_STAMP(b, i, v) do { if (threadIdx.x == 0 && (b) < 2048) air_wgrad_wg_stamps[(b) * 4 + (i)] = (unsigned long long)(v); } while (0)
#define WG_HWID() ((__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 16) | __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11)))
#else
#define WG_STAMP(b, i, v) do { } while (0)
#define WG_HWID() 0
#endif
__device__ __forceinline__ void run_tile_bf16_body(const Table& tab, int block, unsigned short* Img, float* __restrict__ sq_partials,
                                                   int32_t* __restrict__ istate, float* sq_red);
__device__ __forceinline__ void run_tile_bf16(const Table& tab, int block, unsigned short* Img, float* __restrict__ sq_partials,
                                              int32_t* __restrict__ istate, float* sq_red)
{
    WG_STAMP(block, 0, wall_clock64());
    WG_STAMP(block, 2, WG_HWID());
    run_tile_bf16_body(tab, block, Img, sq_partials, istate, sq_red);
    WG_STAMP(block, 1, wall_clock64());
}
__device__ __forceinline__ void run_tile_bf16_body(const Table& tab, int block, unsigned short* Img, float* __restrict__ sq_partials,
                                                   int32_t* __restrict__ istate, float* sq_red)
{
    if (block < tab.nbias) { run_bias_block(tab, block, sq_partials, sq_red); return; }
    int m0, n0;
    const Prob& pr = find_tile(tab, block, m0, n0);
    WG_STAMP(block, 3, pr.K * 16 + pr.strip);
    AIR_STAMP(0);
    if (pr.strip > 0) {                      // K = 64 / 128 / 192 / 256 (strip_of)
        const bool a16 = (pr.lda & 7) == 0 && (reinterpret_cast<uintptr_t>(pr.A16) & 15) == 0 && (pr.M % BT) == 0;
        if (a16) {
            if (pr.K == 256) run_strip_bf16<8, false>(pr, m0, n0, Img, sq_partials, istate, sq_red);
            else if (pr.K == 192) run_strip_bf16<6, false>(pr, m0, n0, Img, sq_partials, istate, sq_red);
            else if (pr.K == 128) run_strip_bf16<4, false>(pr, m0, n0, Img, sq_partials, istate, sq_red);
            else run_strip_bf16<2, false>(pr, m0, n0, Img, sq_partials, istate, sq_red);
        } else {
            if (pr.K == 256) run_strip_bf16<8, true>(pr, m0, n0, Img, sq_partials, istate, sq_red);
            else if (pr.K == 192) run_strip_bf16<6, true>(pr, m0, n0, Img, sq_partials, istate, sq_red);
            else if (pr.K == 128) run_strip_bf16<4, true>(pr, m0, n0, Img, sq_partials, istate, sq_red);
            else run_strip_bf16<2, true>(pr, m0, n0, Img, sq_partials, istate, sq_red);
        }
        return;
    }
    // block-uniform: operands from their bf16 twins where the problem supplies usable ones
    const float bias_sq = twin_ok(pr) ? tile_bf16_tw<NIMG_W>(pr, m0, n0, Img) : tile_bf16<NIMG_W>(pr, m0, n0, Img);
    AIR_STAMP(6);
    const float sq = store_tile(pr, m0, n0, reinterpret_cast<const float*>(Img), bias_sq);
    AIR_STAMP(7);
    if (sq_partials) publish_sq(sq, sq_partials, istate, pr.first_part + (block - pr.first_block), sq_red);
    AIR_STAMP(8);
}

}  // namespace airw
