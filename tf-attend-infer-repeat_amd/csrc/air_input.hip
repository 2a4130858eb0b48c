// tf.train.shuffle_batch as device work (multi_mnist.py:228-249; training.py:76-81): see air_shuffle_batch_t in air_hip.h.
// One workgroup per dequeue: the `batch` picks of RandomShuffleQueue::TryDequeueMany, resolved in parallel (dequeue_batch),
// the freed back slots refilled from the stream; the batch's rows are then gathered (batch_gather_kernel).
#include "air_common.h"
#include "air_philox.h"

constexpr int SQ_THREADS = 1024;

__global__ __launch_bounds__(SQ_THREADS) void shuffle_init_kernel(air_shuffle_batch_t a) {
    for (int i = threadIdx.x; i < a.capacity; i += SQ_THREADS) a.queue[i] = i % a.n_records;
    if (threadIdx.x == 0) { a.state[0] = a.capacity; a.state[1] = 0; }
}

// One dequeue of `batch` elements, by all threads of the workgroup, on the queue image `q` (LDS or memory).
// RandomShuffleQueue::TryDequeueMany makes the picks one after the other -- pick k: idx_k = r_k mod (capacity - k), emit
// q[idx_k], q[idx_k] = q[capacity - 1 - k] (the back), pop -- a chain of `batch` dependent read-modify-writes (64 LDS round
// trips: the ~6-12 us of the first version of this kernel).  The chain is resolved in parallel instead: idx_k does not
// depend on the earlier picks (the size at pick k is capacity - k), and the content of a slot at the time of pick k is the
// ORIGINAL content of the slot found by walking the earlier picks backwards -- V(s, t): the last j < t with idx_j == s moved
// the back of that time into s, so V(s, t) = V(capacity - 1 - j, j); no such j: q0[s].  t only decreases, so ONE descending
// pass over j resolves it.  Thread k resolves V(idx_k, k) (what it emits) and V(back_k, k) (what it moves into idx_k); the
// last pick that hits a surviving slot writes it; the `batch` freed slots at the back are refilled from the stream.  Pick
// for pick the sequential queue (tests/test_shuffle_queue.py: 400 batches x 3 geometries against the numpy model).
__device__ __forceinline__ void dequeue_batch(int* q, const air_shuffle_batch_t& a, long n, unsigned pos_mod /* stream position mod n_records */,
                                              int* sh_idx, unsigned* sh_bits /* [capacity / 32 + 1], all zero on entry and on exit */,
                                              int32_t* picks_out) {
    const int tid = threadIdx.x, cap = a.capacity, batch = a.batch;
    if (tid * 4 < batch) {
        uint32_t c[4] = {(uint32_t)n, (uint32_t)((unsigned long)n >> 32), (uint32_t)tid, 0x53485546u};
        air_philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
#pragma unroll
        for (int k = 0; k < 4; ++k) sh_idx[tid * 4 + k] = (int)(c[k] % (uint32_t)(cap - (tid * 4 + k)));
    }
    __syncthreads();
    const int k = tid;
    int my = -1, s = 0, sb = 0, emit = 0, moved = 0;
    bool last_writer = true;
    if (batch <= 64) {
        // the reference's batch: the picks live in the lanes of ONE wave.  A pick is touched by the earlier picks of its
        // batch only through an EVENT: two picks on one slot, or a pick that lands among the last `batch` slots (the backs
        // that move).  Both are detected exactly -- a bitmap of the picked slots in LDS, one atomic OR per lane -- and a
        // batch without events (more than half of them at the reference's geometry) needs no walk at all: pick k emits
        // q[idx_k] and moves q[capacity - 1 - k].  Otherwise the wave walks the picks once, branch-free, reading them by
        // v_readlane (a walk with per-lane branches was 35 instructions per step: 5 of the 6 us of a dequeue).
        if (tid < 64) {
            const bool live = k < batch;
            my = live ? sh_idx[k] : -1;
            s = my;                   // -> the original slot whose content pick k emits
            sb = cap - 1 - k;         // -> the original slot whose content pick k moves into idx_k
            bool event = false;
            unsigned bit = 0;
            if (live) {
                bit = 1u << (my & 31);
                event = (atomicOr(&sh_bits[my >> 5], bit) & bit) != 0 || my >= cap - batch;
            }
            if (__any(event)) {
#pragma unroll 4
                for (int j = batch - 1; j >= 0; --j) {               // uniform loop: j is a scalar
                    const int ij = __builtin_amdgcn_readlane(my, j), bj = cap - 1 - j;
                    const bool before = j < k;
                    last_writer = last_writer && !(j > k && ij == my);
                    s = (before && ij == s) ? bj : s;
                    sb = (before && ij == sb) ? bj : sb;
                }
            }
            if (live) {
                emit = q[s];
                moved = q[sb];
                atomicAnd(&sh_bits[my >> 5], ~bit);                  // the bitmap is all zero again
            }
        }
    } else if (k < batch) {
        my = sh_idx[k];
        s = my;
        sb = cap - 1 - k;
        for (int j = batch - 1; j > k; --j) last_writer = last_writer && sh_idx[j] != my;
        for (int j = k - 1; j >= 0; --j) {
            const int ij = sh_idx[j], bj = cap - 1 - j;
            if (ij == s) s = bj;
            if (ij == sb) sb = bj;
        }
        emit = q[s];
        moved = q[sb];
    }
    __syncthreads();                  // every read of the old image before any write
    if (k < batch) {
        picks_out[k] = emit;
        if (last_writer && my < cap - batch) q[my] = moved;
        q[cap - batch + k] = (int)((pos_mod + (unsigned)k) % (unsigned)a.n_records);   // enqueue appends at the back, in stream order
    }
    __syncthreads();
}

constexpr int SQ_BITS_WORDS = 48 * 1024 / 4 / 32 + 1;                  // one bit per slot of the largest queue sq_check admits

__global__ __launch_bounds__(SQ_THREADS) void shuffle_dequeue_kernel(air_shuffle_batch_t a) {
    __shared__ int sh_idx[SQ_THREADS];
    __shared__ unsigned sh_bits[SQ_BITS_WORDS];
    for (int i = threadIdx.x; i < SQ_BITS_WORDS; i += SQ_THREADS) sh_bits[i] = 0;
    const long pos = a.state[0], n = a.state[1];
    dequeue_batch(a.queue, a, n, (unsigned)(pos % a.n_records), sh_idx, sh_bits, a.picks);   // on the queue in memory: 2 x batch scattered reads, <= 2 x batch writes
    if (threadIdx.x == 0) { a.state[0] = pos + a.batch; a.state[1] = n + 1; }
}

// `nb` dequeues in ONE launch (the picks of a whole hipGraph replay): the queue is staged in LDS once, every batch draws
// with its own dequeue number -- pick for pick what `nb` calls of shuffle_dequeue_kernel make.  picks_out[k * batch + i] =
// pick i of batch k.
__global__ __launch_bounds__(SQ_THREADS) void shuffle_dequeue_many_kernel(air_shuffle_batch_t a, int nb, int32_t* __restrict__ picks_out) {
    extern __shared__ int sq_lds[];
    int* q = sq_lds;                                                   // [capacity]
    int* sh_idx = sq_lds + a.capacity;                                 // [batch]
    __shared__ unsigned sh_bits[SQ_BITS_WORDS];
    const int tid = threadIdx.x;
    for (int i = tid; i < SQ_BITS_WORDS; i += SQ_THREADS) sh_bits[i] = 0;
    const long pos0 = a.state[0], n0 = a.state[1];
    for (int i = tid; i < a.capacity; i += SQ_THREADS) q[i] = a.queue[i];
    __syncthreads();
    unsigned pm = (unsigned)(pos0 % a.n_records);                      // (one 64-bit division per launch, not one per record)
    for (int kb = 0; kb < nb; ++kb) {
        dequeue_batch(q, a, n0 + kb, pm, sh_idx, sh_bits, picks_out + (size_t)kb * a.batch);
        pm = (pm + (unsigned)a.batch) % (unsigned)a.n_records;
    }
    for (int i = tid; i < a.capacity; i += SQ_THREADS) a.queue[i] = q[i];
    if (tid == 0) { a.state[0] = pos0 + (long)nb * a.batch; a.state[1] = n0 + nb; }
}

// the batch's rows: out_images[i] = images[picks[i]], out_digits[i] = digits[picks[i]] (read_and_decode's tensors after the
// queue, multi_mnist.py:228-249).  One workgroup per (row, 1024-float4 slice); D % 4 == 0.
constexpr int GB_THREADS = 256;
__global__ __launch_bounds__(GB_THREADS) void batch_gather_kernel(const float* __restrict__ images, const int32_t* __restrict__ digits,
                                                                  const int32_t* __restrict__ picks, float* __restrict__ out_images,
                                                                  int32_t* __restrict__ out_digits, int D4) {
    const int row = blockIdx.x, rec = picks[row];
    const float4* src = reinterpret_cast<const float4*>(images) + (size_t)rec * D4;
    float4* dst = reinterpret_cast<float4*>(out_images) + (size_t)row * D4;
    for (int i = blockIdx.y * GB_THREADS + threadIdx.x; i < D4; i += gridDim.y * GB_THREADS) dst[i] = src[i];
    if (blockIdx.y == 0 && threadIdx.x == 0 && out_digits) out_digits[row] = digits[rec];
}

static int sq_check(const air_shuffle_batch_t* a) {
    if (!a || !a->queue || !a->state || !a->picks) return AIR_EINVAL;
    if (a->capacity <= 0 || a->batch <= 0 || a->n_records <= 0 || a->min_after_dequeue < 0) return AIR_EINVAL;
    if (a->batch % 4 || a->batch > SQ_THREADS) return AIR_EALIGN;
    if (a->capacity - a->batch < a->min_after_dequeue) return AIR_EINVAL;
    if ((size_t)a->capacity * 4 + (size_t)a->batch * 8 > 48 * 1024) return AIR_ELIMIT;
    return 0;
}

extern "C" int air_shuffle_batch_init(const air_shuffle_batch_t* a, void* stream) {
    if (int rc = sq_check(a)) return rc;
    hipLaunchKernelGGL(shuffle_init_kernel, dim3(1), dim3(SQ_THREADS), 0, air_stream(stream), *a);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_shuffle_batch_dequeue(const air_shuffle_batch_t* a, void* stream) {
    if (int rc = sq_check(a)) return rc;
    hipLaunchKernelGGL(shuffle_dequeue_kernel, dim3(1), dim3(SQ_THREADS), 0, air_stream(stream), *a);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_shuffle_batch_dequeue_many(const air_shuffle_batch_t* a, int n_batches, int32_t* picks_out, void* stream) {
    if (int rc = sq_check(a)) return rc;
    if (n_batches <= 0 || !picks_out) return AIR_EINVAL;
    const size_t lds = (size_t)a->capacity * 4 + (size_t)a->batch * 4;
    hipLaunchKernelGGL(shuffle_dequeue_many_kernel, dim3(1), dim3(SQ_THREADS), lds, air_stream(stream), *a, n_batches, picks_out);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_batch_gather(const float* images, const int32_t* digits, const int32_t* picks, float* out_images,
                                int32_t* out_digits, int batch, int D, void* stream) {
    if (!images || !picks || !out_images || batch <= 0 || D <= 0) return AIR_EINVAL;
    if (out_digits && !digits) return AIR_EINVAL;
    if (D % 4 || ((uintptr_t)images | (uintptr_t)out_images) % 16) return AIR_EALIGN;
    const int D4 = D / 4, slices = (D4 + 4 * GB_THREADS - 1) / (4 * GB_THREADS);
    hipLaunchKernelGGL(batch_gather_kernel, dim3(batch, slices), dim3(GB_THREADS), 0, air_stream(stream),
                       images, digits, picks, out_images, out_digits, D4);
    AIR_CHECK_LAUNCH();
    return 0;
}
