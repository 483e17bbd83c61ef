// Diagnostic: MFMA issue rate and shader clock under load.  clock64() = s_memtime, wall_clock64() = 100 MHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(int iters, unsigned long long* out, float* sink) {
    bf16x8 av[4], bw[4];
    unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (int r = 0; r < 4; ++r)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u; av[r][i] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 0.001f);
            h = h * 1664525u + 1013904223u; bw[r][i] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 0.001f);
        }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i & 3], bw[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    __shared__ unsigned long long smin, smax;
    if (threadIdx.x == 0) { smin = ~0ull; smax = 0; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicMin(&smin, c0); atomicMax(&smax, c1); }
    __syncthreads();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = smax - smin; out[blockIdx.x * 2 + 1] = w1 - w0; }
}
int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 256, iters = argc > 3 ? atoi(argv[3]) : 200000;
    unsigned long long* d; float* sink;
    hipMalloc(&d, blocks * 16); hipMalloc(&sink, (size_t)blocks * threads * 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, iters, d, sink);
        hipDeviceSynchronize();
    }
    unsigned long long* h = (unsigned long long*)malloc(blocks * 16);
    hipMemcpy(h, d, blocks * 16, hipMemcpyDeviceToHost);
    double c = 0, w = 0;
    for (int i = 0; i < blocks; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
    c /= blocks; w /= blocks;
    double mf = (double)iters * 8;
    printf("blocks=%d threads=%d: block span ticks per (MFMA per wave) %.2f, wall(100MHz) ns/MFMA %.3f, memtime rate %.1f MHz, per-SIMD TFLOP-equiv chip: %.0f TF\n",
           blocks, threads, c / mf, w * 10 / mf, c / (w * 10) * 1000,
           (double)blocks * (threads / 64) * mf * 16384 / (w * 10e-9) / 1e12);
    return 0;
}
