// Diagnostic: one 32-K step of a 256x256 GEMM tile done by FOUR waves (128x128 each, one wave per SIMD, 512-register
// budget) instead of eight (128x64, two per SIMD): 64 MFMA 16x16x32 per wave per step, 16 ds_read_b128, 8 global_load_lds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int BAR, int LDSR, int GLDS>
__global__ __launch_bounds__(256) void k(int steps, const char* src, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, frow = lane & 15, fq = lane >> 4;
    const int coff = (fq ^ (((frow >> 3) & 1) << 1)) << 4;
    const int x_off = (wm * 128 + frow) * 64 + coff, w_off = 16384 + (wn * 128 + frow) * 64 + coff;
    for (int i = tid; i < 163840 / 4; i += 256) ((float*)smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    bf16x8 wfa[8], wfb[8], xf[8];
    for (int t = 0; t < 8; ++t) wfa[t] = *(const bf16x8*)(smem + w_off + t * 1024);
    for (int t = 0; t < 8; ++t) xf[t] = *(const bf16x8*)(smem + x_off + t * 1024);
    f32x4 acc[8][8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const char* gp = src + ((size_t)blockIdx.x * 256 + tid) * 16;
    __syncthreads();
    unsigned long long c0 = clock64();
    int slot = 0;
    auto step = [&](int s, bf16x8 (&wf)[8], bf16x8 (&wn_)[8]) {
        if (BAR) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(2)\n\ts_barrier" ::: "memory");
        const char* sb = smem + slot * 32768;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
            if (GLDS)
                __builtin_amdgcn_global_load_lds(GLB_PTR(gp + (size_t)((s & 63) * 8 + mt) * 4096 * 256), LDS_PTR(smem + ((slot + 3) % 5) * 32768 + mt * 4096 + wave * 1024), 16, 0, 0);
            if (LDSR) {
                wn_[mt] = *(const bf16x8*)(sb + w_off + mt * 1024);
                xf[mt] = *(const bf16x8*)(sb + x_off + mt * 1024);
            }
        }
        if (LDSR || GLDS) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                if (GLDS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (LDSR) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        }
        slot = slot + 1 == 5 ? 0 : slot + 1;
    };
    for (int s = 0; s < steps; s += 2) { step(s, wfa, wfb); step(s + 1, wfb, wfa); }
    unsigned long long c1 = clock64();
    __syncthreads();
    unsigned long long& smin = *(unsigned long long*)smem;
    unsigned long long& smax = *(unsigned long long*)(smem + 8);
    if (tid == 0) { smin = ~0ull; smax = 0; }
    __syncthreads();
    if (lane == 0) { atomicMin(&smin, c0); atomicMax(&smax, c1); }
    __syncthreads();
    float sum = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][3];
    sink[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) out[blockIdx.x] = smax - smin;
}

template <int BAR, int LDSR, int GLDS>
void run(const char* name, int blocks, int steps, const char* src, unsigned long long* d, float* sink) {
    auto kern = k<BAR, LDSR, GLDS>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 163840, 0, steps, src, d, sink); (void)hipDeviceSynchronize(); }
    unsigned long long* h = (unsigned long long*)malloc(blocks * 8);
    (void)hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < blocks; ++i) c += h[i];
    printf("  %-52s %8.1f cycles / step (ideal 64 MFMA x 16 = 1024)\n", name, c / blocks / steps);
    free(h);
}

int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 8, steps = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned long long* d; float* sink; char* src;
    (void)hipMalloc(&d, blocks * 8); (void)hipMalloc(&sink, (size_t)blocks * 256 * 4);
    size_t srcb = (size_t)512 * 4096 * 256 + (size_t)blocks * 4096 + 65536;
    (void)hipMalloc(&src, srcb); (void)hipMemset(src, 0, srcb);
    printf("4 waves, blocks=%d steps=%d\n", blocks, steps);
    run<0, 0, 0>("MFMA only", blocks, steps, src, d, sink);
    run<1, 0, 0>("+ barrier", blocks, steps, src, d, sink);
    run<1, 1, 0>("+ barrier + 16 rolling ds_read", blocks, steps, src, d, sink);
    run<1, 0, 1>("+ barrier + 8 interleaved global_load_lds", blocks, steps, src, d, sink);
    run<1, 1, 1>("+ barrier + rolling ds_read + global_load_lds", blocks, steps, src, d, sink);
    return 0;
}
