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
