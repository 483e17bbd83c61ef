// Large-tile variant of the bf16 MFMA GEMM (same contract as asrk_gemm.hip) for the big
// activations x weights products of the encoder: 256(m) x 256(n) x 64(k) tiles, 512 threads =
// 8 waves as 2(m) x 4(n), each wave 128 x 64 = 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16
// (128 accumulator registers).  One workgroup per CU (128 KiB LDS = 2 stages x (X 32 KiB + W 32 KiB)).
//
// Pipeline: global_load_lds prefetch runs TWO K tiles ahead and stays in flight across the
// barriers: the loop uses raw s_barrier with counted s_waitcnt vmcnt(8) (never a draining
// __syncthreads), so the HBM/L2 latency of tile t+1 hides under the 64 MFMAs per wave of tile t.
//   top:    vmcnt(8) [my loads of tile t landed]  ; s_barrier  [everyone's landed]
//   body:   24 ds_read_b128 + 64 MFMA per wave from stage t&1
//   bottom: lgkmcnt(0) ; s_barrier [everyone done reading stage t&1] ; issue loads of tile t+2 into it
// LDS image and swizzle are those of asrk_gemm.hip (linear per wave, chunk ^= row & 7 applied to
// the source address and to the fragment read).
#include "asrk_common.h"
#include "asrk_gemm.h"
#include "asrk_prof.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int OPER_BYTES = BM * BK * 2;      // 32 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * OPER_BYTES;  // 64 KiB

template <bool GELU, bool OUT_BF16>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][X tile | W tile]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int tiles_n = (a.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int m0 = (lin / tiles_n) * BM;
    const int n0 = (lin % tiles_n) * BN;

    const int srow = tid >> 3;                 // + 64 * i
    const int chunk = (tid & 7) ^ (srow & 7);  // source-side swizzle
    const bf16_t* xp[4];
    const bf16_t* wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + srow + 64 * i;
        m = m < a.M ? m : a.M - 1;
        xp[i] = a.X + (int64_t)(m / a.x_rpb) * a.x_bs + (int64_t)(m % a.x_rpb) * a.x_rs + chunk * 8;
        int n = n0 + srow + 64 * i;
        n = n < a.N ? n : a.N - 1;
        wp[i] = a.W + (int64_t)n * a.K + chunk * 8;
    }
    const int wave_lds = wave * 1024;

    auto stage = [&](int kt, char* sb) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(xp[i] + k0, sb + i * 8192 + wave_lds);
            glds16(wp[i] + k0, sb + OPER_BYTES + i * 8192 + wave_lds);
        }
    };

    f32x4 acc[4][8];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fq = lane >> 4;
    const int sw = lane & 7;
    const int x_off = (wm * 128 + frow) * 128;
    const int w_off = OPER_BYTES + (wn * 64 + frow) * 128;

    auto compute = [&](const char* sb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + fq) ^ sw) << 4;
            bf16x8 wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(sb + w_off + t * 2048 + coff);
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                bf16x8 xf[4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    xf[t] = *reinterpret_cast<const bf16x8*>(sb + x_off + (mh * 4 + t) * 2048 + coff);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[nt][mh * 4 + t] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[t], acc[nt][mh * 4 + t], 0, 0, 0);
            }
        }
    };

    const int nkt = a.K / BK;
    char* const s0 = smem;
    char* const s1 = smem + STAGE_BYTES;
    stage(0, s0);
    if (nkt > 1) stage(1, s1);
    for (int kt = 0; kt < nkt; kt += 2) {
        // ---- even tile from stage 0
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        compute(s0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < nkt) stage(kt + 2, s0);
        if (kt + 1 >= nkt) break;
        // ---- odd tile from stage 1
        if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        compute(s1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 3 < nkt) stage(kt + 3, s1);
    }

    // ---- epilogue: lane holds 4 consecutive n of row m per accumulator
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int m = m0 + wm * 128 + mt * 16 + frow;
        if (m >= a.M) continue;
        const int64_t orow = (int64_t)(m / a.o_rpb) * a.o_bs + (int64_t)(m % a.o_rpb) * a.o_rs + a.o_off;
        const float* rrow = a.res ? a.res + (int64_t)(m % a.res_rows) * a.ldr : nullptr;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + fq * 4;
            if (n >= a.N) continue;
            gemm_epilogue_store<GELU, OUT_BF16>(a, acc[nt][mt], m, n, orow, rrow);
        }
    }
}

template <bool GELU, bool OUT_BF16>
int launch(const GemmArgs& a, hipStream_t st) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const size_t lds = 2 * STAGE_BYTES;
    auto kern = gemm256_kernel<GELU, OUT_BF16>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ASRK_ERR_HIP;
    AsrkProfScope prof(ASRK_PROF_GEMM, 2.0 * a.M * a.N * a.K, st);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, st, a);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}

}  // namespace

// Arguments are validated by asrk_gemm_launch (asrk_gemm.hip), which dispatches here.
int asrk_gemm256_launch(const GemmArgs& a, bool gelu, bool out_bf16, hipStream_t st) {
    if (a.splits != 1) return ASRK_ERR_SHAPE;
    if (gelu) return out_bf16 ? launch<true, true>(a, st) : launch<true, false>(a, st);
    return out_bf16 ? launch<false, true>(a, st) : launch<false, false>(a, st);
}
