// Experiment: semantics and bank behaviour of gfx950's ds_read_b64_tr_b16 (LDS transpose read), the
// instruction that lets an MFMA B-fragment (8 consecutive k per lane) be read from an LDS image that is
// n-contiguous (rows = k), i.e. a row-major [K,N] weight / a K-slow weight-gradient operand copied
// into LDS as it lies in memory.
//   hipcc --offload-arch=gfx950 -O3 -o tr_read tr_read.hip && ./tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(const unsigned* addr, unsigned short* out, int nelem) {
    extern __shared__ unsigned short lds[];
    for (int i = threadIdx.x; i < nelem; i += blockDim.x) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)lds + addr[threadIdx.x];
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = r.x & 0xffff; out[threadIdx.x * 4 + 1] = r.x >> 16;
    out[threadIdx.x * 4 + 2] = r.y & 0xffff; out[threadIdx.x * 4 + 3] = r.y >> 16;
}

// timing: `reps` reads per lane with per-lane base address + immediate-free walking (same address each time:
// the LDS pipe cost per instruction incl. bank conflicts)
template <int MODE>
__global__ void timing(const unsigned* addr, unsigned* sink, long long* cycles, int reps) {
    extern __shared__ unsigned short lds[];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)lds + addr[threadIdx.x];
    unsigned acc = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int k = 0; k < reps; ++k) {
        if (MODE == 0) {
            uint2 r0, r1, r2, r3;
            asm volatile("ds_read_b64_tr_b16 %0, %4\n ds_read_b64_tr_b16 %1, %4 offset:2048\n"
                         "ds_read_b64_tr_b16 %2, %4 offset:4096\n ds_read_b64_tr_b16 %3, %4 offset:6144\n s_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a) : "memory");
            acc += r0.x + r1.y + r2.x + r3.y;
        } else if (MODE == 1) {
            uint2 r0, r1, r2, r3;
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:2048\n"
                         "ds_read_b64 %2, %4 offset:4096\n ds_read_b64 %3, %4 offset:6144\n s_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a) : "memory");
            acc += r0.x + r1.y + r2.x + r3.y;
        } else {
            uint4 r0, r1, r2, r3;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:2048\n"
                         "ds_read_b128 %2, %4 offset:4096\n ds_read_b128 %3, %4 offset:6144\n s_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a) : "memory");
            acc += r0.x + r1.y + r2.z + r3.w;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    sink[threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

static void run_probe(const char* name, const std::vector<unsigned>& addr) {
    unsigned* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
    hipMemcpy(d_addr, addr.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 65536, 0, d_addr, d_out, 32768);
    std::vector<unsigned short> out(256);
    hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost);
    printf("== %s: lane: addr(elem) -> 4 elements read\n", name);
    for (int l = 0; l < 64; ++l)
        printf("  lane %2d addr %5u -> %5u %5u %5u %5u\n", l, addr[l] / 2, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
    hipFree(d_addr); hipFree(d_out);
}

template <int MODE>
static double run_timing(const std::vector<unsigned>& addr, int threads) {
    unsigned* d_addr; unsigned* d_sink; long long* d_cyc;
    hipMalloc(&d_addr, threads * 4); hipMalloc(&d_sink, threads * 4); hipMalloc(&d_cyc, 8);
    hipMemcpy(d_addr, addr.data(), threads * 4, hipMemcpyHostToDevice);
    const int reps = 2000;
    hipLaunchKernelGGL(timing<MODE>, dim3(1), dim3(threads), 65536, 0, d_addr, d_sink, d_cyc, reps);
    hipLaunchKernelGGL(timing<MODE>, dim3(1), dim3(threads), 65536, 0, d_addr, d_sink, d_cyc, reps);
    long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
    hipFree(d_addr); hipFree(d_sink); hipFree(d_cyc);
    return (double)c / (reps * 4);
}

int main() {
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)timing<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)timing<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)timing<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    std::vector<unsigned> a(64);
    // (1) lane l reads the 8 bytes at l*8: result element values identify (source lane, element)
    for (int l = 0; l < 64; ++l) a[l] = l * 8;
    run_probe("linear 8 B per lane", a);
    // (2) the layout an MFMA 16x16x32 B-fragment wants from a [k][16 n] image (32 B per k row):
    //     group g = l>>4 needs k = 8g..8g+3 (first read): lane i of the group -> row 8g + i/4, column quad i%4
    for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; a[l] = ((8 * g + i / 4) * 16 + (i % 4) * 4) * 2; }
    run_probe("[k][16] image, rows 8g+i/4, quad i%4 (expect lane n: k = 8g..8g+3 of column n)", a);
    // timings: wave64, candidate row strides (elements) for an image [k][S]: bank conflicts of the tr read
    printf("== cycles per instruction (4 in flight), one wave\n");
    for (int S : {16, 32, 64, 128, 24, 40, 72, 136, 20, 36, 68}) {
        std::vector<unsigned> t(64);
        for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; t[l] = ((8 * g + i / 4) * S + (i % 4) * 4) * 2; }
        printf("  tr_b16  [k][%3d] rows 8g+i/4: %.1f\n", S, run_timing<0>(t, 64));
    }
    for (int S : {16, 32, 64, 128}) {
        std::vector<unsigned> t(64);
        // variant: all four groups read the SAME k rows but different 16-column blocks (wave covers 64 columns)
        for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; t[l] = ((i / 4) * S + g * 16 + (i % 4) * 4) * 2; }
        printf("  tr_b16  [k][%3d] rows i/4, col block g: %.1f\n", S, run_timing<0>(t, 64));
    }
    {
        std::vector<unsigned> t(64);
        for (int l = 0; l < 64; ++l) t[l] = l * 8;
        printf("  tr_b16  linear 8 B/lane: %.1f   ds_read_b64 linear: %.1f\n", run_timing<0>(t, 64), run_timing<1>(t, 64));
        for (int l = 0; l < 64; ++l) t[l] = l * 16;
        printf("  ds_read_b128 linear 16 B/lane: %.1f\n", run_timing<2>(t, 64));
    }
    // four waves issuing concurrently (the GEMM's situation)
    for (int S : {16, 32, 64}) {
        std::vector<unsigned> t(256);
        for (int l = 0; l < 256; ++l) { int w = l >> 6, g = (l >> 4) & 3, i = l & 15; t[l] = (w * 16384) + ((8 * g + i / 4) * S + (i % 4) * 4) * 2; }
        printf("  4 waves tr_b16 [k][%3d]: %.1f per instruction per wave\n", S, run_timing<0>(t, 256));
    }
    return 0;
}
