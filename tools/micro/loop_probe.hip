// Diagnostic: cost of one 32-K step of the 256x256 GEMM loop (8 waves x 32 MFMA 16x16x32) with its parts
// switched on one by one: workgroup barrier, 12 ds_read_b128 per wave, 4 global_load_lds per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int BAR, int LDSR, int GLDS, int ROLL>
__global__ __launch_bounds__(512, 2) void k(int steps, const char* src, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, frow = lane & 15, fq = lane >> 4;
    const int coff = (fq ^ (((frow >> 3) & 1) << 1)) << 4;
    const int x_off = (wm * 128 + frow) * 64 + coff, w_off = 16384 + (wn * 64 + frow) * 64 + coff;
    for (int i = tid; i < 163840 / 4; i += 512) ((float*)smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    bf16x8 wf[4], xf[8];
    for (int t = 0; t < 4; ++t) wf[t] = *(const bf16x8*)(smem + w_off + t * 1024);
    for (int t = 0; t < 8; ++t) xf[t] = *(const bf16x8*)(smem + x_off + t * 1024);
    f32x4 acc[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const char* gp = src + ((size_t)blockIdx.x * 512 + tid) * 16;
    __syncthreads();
    unsigned long long c0 = clock64();
    int slot = 0;
    for (int s = 0; s < steps; ++s) {
        if (BAR) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(2)\n\ts_barrier" ::: "memory");
        const char* sb = smem + slot * 32768;
        if (GLDS == 2) {  // plain loads into registers, consumed by nothing (asm keeps them)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 v = *(const uint4*)(gp + (size_t)((s & 63) * 4 + i) * 8192 * 256);
                asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            }
        }
        if (GLDS == 3) {  // 2 loads of the same total... half the instructions
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds(GLB_PTR(gp + (size_t)((s & 63) * 4 + i) * 8192 * 256), LDS_PTR(smem + ((slot + 3) % 5) * 32768 + i * 8192 + wave * 1024), 16, 0, 0);
        }
        if (GLDS == 4) {  // same address every step (L2/TCP-hot), 4 per step
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds(GLB_PTR(gp + (size_t)i * 8192 * 256), LDS_PTR(smem + ((slot + 3) % 5) * 32768 + i * 8192 + wave * 1024), 16, 0, 0);
        }
        if (GLDS == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds(GLB_PTR(gp + (size_t)((s & 63) * 4 + i) * 8192 * 256), LDS_PTR(smem + ((slot + 3) % 5) * 32768 + i * 8192 + wave * 1024), 16, 0, 0);
        }
        if (ROLL) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
                if (GLDS == 5 && (mt & 1)) {
                    const int i = mt >> 1;
                    __builtin_amdgcn_global_load_lds(GLB_PTR(gp + (size_t)((s & 63) * 4 + i) * 8192 * 256), LDS_PTR(smem + ((slot + 3) % 5) * 32768 + i * 8192 + wave * 1024), 16, 0, 0);
                }
                if (LDSR) {
                    if (mt == 0) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) wf[t] = *(const bf16x8*)(sb + w_off + t * 1024);  // (value reuse is fine for timing)
                    }
                    xf[mt] = *(const bf16x8*)(sb + x_off + mt * 1024);
                }
            }
            if (LDSR) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
                for (int mt = 1; mt < 8; ++mt) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    if (GLDS == 5 && (mt & 1)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        } else {
            if (LDSR) {
#pragma unroll
                for (int t = 0; t < 4; ++t) wf[t] = *(const bf16x8*)(sb + w_off + t * 1024);
#pragma unroll
                for (int t = 0; t < 8; ++t) xf[t] = *(const bf16x8*)(sb + x_off + t * 1024);
            }
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        }
        slot = slot + 1 == 5 ? 0 : slot + 1;
    }
    unsigned long long c1 = clock64();
    __syncthreads();
    unsigned long long& smin = *(unsigned long long*)smem;
    unsigned long long& smax = *(unsigned long long*)(smem + 8);
    if (tid == 0) { smin = ~0ull; smax = 0; }
    __syncthreads();
    if (lane == 0) { atomicMin(&smin, c0); atomicMax(&smax, c1); }
    __syncthreads();
    float sum = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][3];
    sink[blockIdx.x * 512 + tid] = sum;
    if (tid == 0) out[blockIdx.x] = smax - smin;
}

template <int BAR, int LDSR, int GLDS, int ROLL>
void run(const char* name, int blocks, int steps, const char* src, unsigned long long* d, float* sink) {
    auto kern = k<BAR, LDSR, GLDS, ROLL>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 163840, 0, steps, src, d, sink); (void)hipDeviceSynchronize(); }
    unsigned long long* h = (unsigned long long*)malloc(blocks * 8);
    (void)hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < blocks; ++i) c += h[i];
    printf("  %-46s %8.1f cycles / step (ideal 64 MFMA x 16 = 1024)\n", name, c / blocks / steps);
    free(h);
}

int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 8, steps = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned long long* d; float* sink; char* src;
    (void)hipMalloc(&d, blocks * 8); (void)hipMalloc(&sink, (size_t)blocks * 512 * 4);
    size_t srcb = (size_t)256 * 8192 * 256 + (size_t)blocks * 8192;
    (void)hipMalloc(&src, srcb); (void)hipMemset(src, 0, srcb);
    printf("blocks=%d steps=%d\n", blocks, steps);
    run<0, 0, 0, 0>("MFMA only", blocks, steps, src, d, sink);
    run<1, 0, 0, 0>("+ barrier", blocks, steps, src, d, sink);
    run<0, 1, 0, 0>("+ 12 ds_read up front (no barrier)", blocks, steps, src, d, sink);
    run<1, 1, 0, 0>("+ barrier + 12 ds_read up front", blocks, steps, src, d, sink);
    run<0, 1, 0, 1>("+ rolling ds_read (no barrier)", blocks, steps, src, d, sink);
    run<1, 1, 0, 1>("+ barrier + rolling ds_read", blocks, steps, src, d, sink);
    run<1, 0, 1, 0>("+ barrier + 4 global_load_lds", blocks, steps, src, d, sink);
    run<1, 0, 2, 0>("+ barrier + 4 global_load_dwordx4 (registers)", blocks, steps, src, d, sink);
    run<1, 0, 3, 0>("+ barrier + 2 global_load_lds", blocks, steps, src, d, sink);
    run<1, 0, 4, 0>("+ barrier + 4 global_load_lds, fixed address", blocks, steps, src, d, sink);
    run<0, 0, 1, 0>("+ 4 global_load_lds, no barrier", blocks, steps, src, d, sink);
    run<1, 1, 1, 1>("+ barrier + rolling ds_read + global_load_lds", blocks, steps, src, d, sink);
    run<1, 1, 5, 1>("+ barrier + rolling ds_read + interleaved global_load_lds", blocks, steps, src, d, sink);
    run<1, 1, 1, 0>("+ barrier + up-front ds_read + global_load_lds", blocks, steps, src, d, sink);
    return 0;
}
