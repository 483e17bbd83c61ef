// Issue cost of v_exp_f32 / v_add_f32 / v_cvt_pk_bf16_f32 / v_max3_f32 per wave64, alone on a SIMD and with 2-3 waves on it.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a[i]));
            if (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&a[i & ~1]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int waves_per_simd = 1; waves_per_simd <= 3; ++waves_per_simd) {
        k<OP><<<1, 256 * waves_per_simd>>>(out, cyc, iters);
        k<OP><<<1, 256 * waves_per_simd>>>(out, cyc, iters);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-18s %d wave(s)/SIMD: %.2f cycles per instruction per wave, %.2f per SIMD-instruction\n", name, waves_per_simd,
               (double)c / (iters * 16.0), (double)c / (iters * 16.0 * waves_per_simd));
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("v_exp_f32"); run<4>("v_rcp_f32"); run<1>("v_add_f32"); run<2>("v_cvt_pk_bf16_f32"); run<3>("v_max3_f32"); run<5>("v_pk_fma_f32");
    return 0;
}
