// tf.train.shuffle_batch as device work (multi_mnist.py:228-249; training.py:76-81): see air_shuffle_batch_t in air_hip.h.
// One workgroup: the queue (43 KB of record indices at the reference's capacity) is staged in LDS, the `batch` dependent
// picks are made by one lane on LDS words (64 x (one read of the picked slot, one of the back, one write)), the freed back
// slots are refilled from the stream, the queue goes back to HBM.  ~6 us; the batch's rows are then gathered by the caller.
#include "air_common.h"
#include "air_philox.h"

constexpr int SQ_THREADS = 1024;

__global__ __launch_bounds__(SQ_THREADS) void shuffle_init_kernel(air_shuffle_batch_t a) {
    for (int i = threadIdx.x; i < a.capacity; i += SQ_THREADS) a.queue[i] = i % a.n_records;
    if (threadIdx.x == 0) { a.state[0] = a.capacity; a.state[1] = 0; }
}

__global__ __launch_bounds__(SQ_THREADS) void shuffle_dequeue_kernel(air_shuffle_batch_t a) {
    extern __shared__ int sq_lds[];
    int* q = sq_lds;                                                   // [capacity]
    uint32_t* r = reinterpret_cast<uint32_t*>(sq_lds + a.capacity);    // [batch] the draws
    int* out = sq_lds + a.capacity + a.batch;                          // [batch] the picks
    const int tid = threadIdx.x;
    const long pos = a.state[0], n = a.state[1];
    for (int i = tid; i < a.capacity; i += SQ_THREADS) q[i] = a.queue[i];
    if (tid * 4 < a.batch) {
        uint32_t c[4] = {(uint32_t)n, (uint32_t)((unsigned long)n >> 32), (uint32_t)tid, 0x53485546u};
        air_philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
#pragma unroll
        for (int k = 0; k < 4; ++k) r[tid * 4 + k] = c[k];
    }
    __syncthreads();
    if (tid == 0) {
        int size = a.capacity;
        for (int k = 0; k < a.batch; ++k) {                            // RandomShuffleQueue: uniform index, swap with the back, pop
            const int idx = (int)(r[k] % (uint32_t)size);
            out[k] = q[idx];
            q[idx] = q[size - 1];
            --size;
        }
    }
    __syncthreads();
    if (tid < a.batch) {
        a.picks[tid] = out[tid];
        q[a.capacity - a.batch + tid] = (int)((pos + tid) % a.n_records);   // enqueue appends at the back, in stream order
    }
    __syncthreads();
    for (int i = tid; i < a.capacity; i += SQ_THREADS) a.queue[i] = q[i];
    if (tid == 0) { a.state[0] = pos + a.batch; a.state[1] = n + 1; }
}

// `nb` dequeues in ONE launch (the picks of a whole hipGraph replay, made on a side branch while the previous replay's steps
// run): the queue is staged once, every batch draws with its own dequeue number -- pick for pick what `nb` calls of
// shuffle_dequeue_kernel make.  picks_out[k * batch + i] = pick i of batch k.
__global__ __launch_bounds__(SQ_THREADS) void shuffle_dequeue_many_kernel(air_shuffle_batch_t a, int nb, int32_t* __restrict__ picks_out) {
    extern __shared__ int sq_lds[];
    int* q = sq_lds;
    uint32_t* r = reinterpret_cast<uint32_t*>(sq_lds + a.capacity);
    int* out = sq_lds + a.capacity + a.batch;
    const int tid = threadIdx.x;
    const long pos0 = a.state[0], n0 = a.state[1];
    for (int i = tid; i < a.capacity; i += SQ_THREADS) q[i] = a.queue[i];
    for (int kb = 0; kb < nb; ++kb) {
        const long n = n0 + kb, pos = pos0 + (long)kb * a.batch;
        if (tid * 4 < a.batch) {
            uint32_t c[4] = {(uint32_t)n, (uint32_t)((unsigned long)n >> 32), (uint32_t)tid, 0x53485546u};
            air_philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
#pragma unroll
            for (int k = 0; k < 4; ++k) r[tid * 4 + k] = c[k];
        }
        __syncthreads();
        if (tid == 0) {
            int size = a.capacity;
            for (int k = 0; k < a.batch; ++k) {
                const int idx = (int)(r[k] % (uint32_t)size);
                out[k] = q[idx];
                q[idx] = q[size - 1];
                --size;
            }
        }
        __syncthreads();
        if (tid < a.batch) {
            picks_out[(size_t)kb * a.batch + tid] = out[tid];
            q[a.capacity - a.batch + tid] = (int)((pos + tid) % a.n_records);
        }
        __syncthreads();
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
    const size_t lds = (size_t)a->capacity * 4 + (size_t)a->batch * 8;
    hipLaunchKernelGGL(shuffle_dequeue_kernel, dim3(1), dim3(SQ_THREADS), lds, air_stream(stream), *a);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_shuffle_batch_dequeue_many(const air_shuffle_batch_t* a, int n_batches, int32_t* picks_out, void* stream) {
    if (int rc = sq_check(a)) return rc;
    if (n_batches <= 0 || !picks_out) return AIR_EINVAL;
    const size_t lds = (size_t)a->capacity * 4 + (size_t)a->batch * 8;
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
