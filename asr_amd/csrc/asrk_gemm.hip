// bf16 MFMA GEMM with fused epilogues for gfx950:  out[m][n] = epi( sum_k X[m][k] * W[n][k] ).
//
// Every nn.Linear on the hot path (HF modeling_whisper.py:279-282 q/k/v/out_proj, :375-376 fc1/fc2;
// reference src/modeling.py:68-70 linear_1/linear_2) and, through the row maps below, both
// convolutions of the stem (HF modeling_whisper.py:566-567, 618-619) run through this kernel.
//
// Layout: X rows and W rows are both K-contiguous (nn.Linear keeps weight as [out][in]), which
// is exactly what the MFMA A/B fragments want (8 consecutive k per lane), so no transposes.
// Tile: 128(m) x 128(n) x 64(k), 256 threads = 4 waves as 2(n) x 2(m); each wave owns 64 x 64
// = 4 x 4 accumulators of v_mfma_f32_16x16x32_bf16.  W is the MFMA "A" operand and X the "B"
// operand, so a lane's 4 accumulator registers are 4 CONSECUTIVE n of one row m: bias / residual
// are read and the result is stored as one 8-byte (bf16) or 16-byte (f32) vector per lane.
// Staging: global_load_lds_dwordx4 (HBM -> LDS without VGPRs), double buffered, one barrier per
// K tile.  The LDS image is linear per wave (hardware constraint) so the XOR bank swizzle
// chunk ^= row & 7 is applied to the per-lane SOURCE address and again on the fragment read.
// Row maps: X row m lives at (m / x_rpb) * x_bs + (m % x_rpb) * x_rs (elements).  With the input
// stored time-major and zero rows around each clip, a k=3 convolution's im2col row is simply 3
// consecutive time rows = one contiguous K = 3*C span, so conv1/conv2 are plain GEMMs with
// overlapping rows (x_rs = C for stride 1, 2C for stride 2).  The output has the same kind of map.
#include "asrk_common.h"
#include "asrk_gemm.h"
#include "asrk_prof.h"

#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per buffer

template <bool GELU, bool OUT_BF16>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][X tile | W tile]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;

    // ---- XCD-aware tile order: blocks b and b+8 share an XCD (L2); give each XCD a contiguous
    //      run of tiles, n fastest, so the X panel and W stay in that XCD's L2.
    const int tiles_n = (a.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int m0 = (lin / tiles_n) * BM;
    const int n0 = (lin % tiles_n) * BN;

    // ---- per-thread staging sources: 4 X rows and 4 W rows, one 16-byte chunk each per K tile
    const int srow = tid >> 3;                   // + 32 * i
    const int chunk = (tid & 7) ^ (srow & 7);    // source-side swizzle (same for all 4 rows)
    const bf16_t* xp[4];
    const bf16_t* wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + srow + 32 * i;
        m = m < a.M ? m : a.M - 1;
        xp[i] = a.X + (int64_t)(m / a.x_rpb) * a.x_bs + (int64_t)(m % a.x_rpb) * a.x_rs + chunk * 8;
        int n = n0 + srow + 32 * i;
        n = n < a.N ? n : a.N - 1;
        wp[i] = a.W + (int64_t)n * a.K + chunk * 8;
    }
    const int wave_lds = wave * 1024;  // this wave's 64 x 16 B slice of each 256-lane round

    auto stage = [&](int kt, int buf) {
        char* xb = smem + buf * (2 * TILE_BYTES);
        char* wb = xb + TILE_BYTES;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(xp[i] + k0, xb + i * 4096 + wave_lds);
            glds16(wp[i] + k0, wb + i * 4096 + wave_lds);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row = base + (lane & 15), chunk c = ks*4 + (lane >> 4), swizzled by row & 7
    const int frow = lane & 15;
    const int fq = lane >> 4;
    const int sw = lane & 7;
    const int w_off = (wn * 64 + frow) * 128;
    const int x_off = (wm * 64 + frow) * 128;

    const int nkt_all = a.K / BK;
    const int per_split = (nkt_all + a.splits - 1) / a.splits;
    const int kt_begin = blockIdx.y * per_split;
    const int nkt = min(per_split, nkt_all - kt_begin);  // may be <= 0 for a trailing split: writes zeros
    if (nkt > 0) stage(kt_begin, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // tile kt has landed for every wave; everyone finished reading the other buffer
        if (kt + 1 < nkt) stage(kt_begin + kt + 1, (kt + 1) & 1);
        const char* xb = smem + (kt & 1) * (2 * TILE_BYTES);
        const char* wb = xb + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + fq) ^ sw) << 4;
            bf16x8 wf[4], xf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                wf[t] = *reinterpret_cast<const bf16x8*>(wb + w_off + t * 2048 + coff);
                xf[t] = *reinterpret_cast<const bf16x8*>(xb + x_off + t * 2048 + coff);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        }
    }

    // ---- epilogue: lane holds n = nb + 0..3 (4 consecutive) for row m
    f32x4 bv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wn * 64 + nt * 16 + fq * 4;
        bv[nt] = (a.bias && n < a.N) ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + frow;
        if (m >= a.M) continue;
        const int64_t orow = (int64_t)(m / a.o_rpb) * a.o_bs + (int64_t)(m % a.o_rpb) * a.o_rs + a.o_off +
                             (int64_t)blockIdx.y * a.o_split;
        const float* rrow = a.res ? a.res + (int64_t)(m % a.res_rows) * a.ldr : nullptr;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + fq * 4;
            if (n >= a.N) continue;
            gemm_epilogue_store<GELU, OUT_BF16>(a, acc[nt][mt], bv[nt], m, n, orow, rrow);
        }
    }
}

template <bool GELU, bool OUT_BF16>
int launch(const GemmArgs& a, hipStream_t st) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const size_t lds = 4 * TILE_BYTES;
    auto kern = gemm_bf16_kernel<GELU, OUT_BF16>;
    static std::atomic<uint64_t> lds_set{0};
    if (asrk_max_lds_once((const void*)kern, (int)lds, lds_set) != ASRK_OK) return ASRK_ERR_HIP;
    AsrkProfScope prof(ASRK_PROF_GEMM128, 2.0 * a.M * a.N * a.K, st);
    hipLaunchKernelGGL(kern, dim3(tiles, a.splits), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}

}  // namespace

int asrk_gemm_launch(const GemmArgs& a, bool gelu, bool out_bf16, hipStream_t st) {
    if (!a.X || !a.W || !a.out) return ASRK_ERR_NULL;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % BK) != 0 || (a.N % 4) != 0) return ASRK_ERR_SHAPE;
    if (a.x_rpb <= 0 || a.o_rpb <= 0 || (a.res && a.res_rows <= 0)) return ASRK_ERR_SHAPE;
    if (a.splits < 1 || a.splits > 1024 || (a.splits > 1 && (a.bias || a.res || a.dg_pre || gelu))) return ASRK_ERR_SHAPE;
    if (((uintptr_t)a.dg_pre & 7) || (a.ld_dg % 4) || (a.o_split % 4)) return ASRK_ERR_ALIGN;
    if (((uintptr_t)a.X & 15) || ((uintptr_t)a.W & 15) || ((uintptr_t)a.out & 15) || ((uintptr_t)a.bias & 15) ||
        ((uintptr_t)a.res & 15))
        return ASRK_ERR_ALIGN;
    if ((a.x_bs % 8) || (a.x_rs % 8) || (a.o_bs % 4) || (a.o_rs % 4) || (a.o_off % 4) || (a.ldr % 4))
        return ASRK_ERR_ALIGN;
    // Tile choice: 256 x 256 tiles (one workgroup per CU) once there are enough of them, 128 x 128 tiles otherwise.
    // Measured crossovers (tools/gemm_bench.py, M swept from 1500 to 96000): the persistent bf16-output kernel wins
    // from about half a tile per CU (141 tiles: 22 us against 28 us; 72 tiles lose), the one-tile-per-workgroup
    // fp32-output kernel only with several rounds of tiles.
#ifdef ASRK_DIAG  // diagnostic build only (libasrk_diag.so): ASRK_GEMM_TILE=128|256 forces the tile size
    static const int forced = [] { const char* e = getenv("ASRK_GEMM_TILE"); return e ? atoi(e) : 0; }();
#else
    constexpr int forced = 0;
#endif
    const long tiles256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    const long min_tiles = out_bf16 ? 136 : 448;  // fp32: large-v3's conv2 at 16 clips (470 tiles) measured 0.274 vs 0.296 ms on the 128-tile kernel
    // a ragged last n-tile is fine while at least 7/8 of the tiles' columns are real (N = 960: 93.75 %)
    const long n_cover = (long)((a.N + 255) / 256) * 256;
    // the 256-tile kernels have no residual add in their bf16 epilogues (no caller on the path combines the two: the
    // residual stream is fp32); that combination stays on the 128-tile kernel, whose epilogue adds it
    const bool big = a.splits == 1 && (forced == 256 || (forced == 0 && tiles256 >= min_tiles && 8L * a.N >= 7L * n_cover)) &&
                     (a.N % 8) == 0 && !(a.res && out_bf16);
    if (big) return asrk_gemm256_launch(a, gelu, out_bf16, st);
    if (gelu) return out_bf16 ? launch<true, true>(a, st) : launch<true, false>(a, st);
    return out_bf16 ? launch<false, true>(a, st) : launch<false, false>(a, st);
}

extern "C" int asrk_gemm_bf16(const void* x, int64_t ldx, const void* w, const float* bias, const float* res,
                              int64_t ldr, int res_rows, void* out, int64_t ldo, int out_dtype, int M, int N, int K,
                              int gelu, float alpha, void* stream) {
    if (out_dtype != ASRK_F32 && out_dtype != ASRK_BF16) return ASRK_ERR_UNSUPPORTED;
    GemmArgs a{};
    a.X = (const bf16_t*)x;
    a.W = (const bf16_t*)w;
    a.bias = bias;
    a.res = res;
    a.out = out;
    a.M = M; a.N = N; a.K = K;
    a.x_rpb = M > 0 ? M : 1; a.x_bs = 0; a.x_rs = ldx;
    a.o_rpb = M > 0 ? M : 1; a.o_bs = 0; a.o_rs = ldo; a.o_off = 0;
    a.ldr = ldr;
    a.res_rows = res ? (res_rows > 0 ? res_rows : M) : 1;
    a.alpha = alpha;
    a.splits = 1;
    if (ldx < K || ldo < N) return ASRK_ERR_SHAPE;
    return asrk_gemm_launch(a, gelu != 0, out_dtype == ASRK_BF16, (hipStream_t)stream);
}
