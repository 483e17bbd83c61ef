// Large-tile variant of the bf16 MFMA GEMM (same contract as asrk_gemm.hip) for the big
// activations x weights products of the encoder: 256(m) x 256(n) tiles, 512 threads = 8 waves as
// 2(m) x 4(n), each wave 128 x 64 = 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16 (128 accumulator
// registers).  One workgroup per CU.
//
// Pipeline: an NSTAGE-deep LDS ring of K slices (BK = 64 x 2 stages, or BK = 32 x 4/5 stages) filled
// by global_load_lds; loads run NSTAGE-1 slices ahead and stay in flight across the barriers: the
// loop uses ONE raw s_barrier per slice with a counted s_waitcnt vmcnt(N) (never a draining
// __syncthreads), so HBM/L2 latency hides under the MFMAs of the slices in between.
//   top of slice t:  vmcnt((NSTAGE-2) * G)  [my loads of slice t landed] ; s_barrier [everyone's landed,
//                    and everyone has finished reading slice t-1]
//                    issue loads of slice t+NSTAGE-1 into the ring slot slice t-1 just vacated
//                    ds_read_b128 fragments + MFMAs of slice t
// LDS image: linear per wave (hardware constraint of LDS-DMA); the XOR bank swizzle is applied to the
// per-lane SOURCE address and again on the fragment read (128-B rows: chunk ^= row & 7; 64-B rows:
// chunk ^= ((row >> 3) & 1) << 1; both conflict-free for ds_read_b128, checked with
// SQ_LDS_BANK_CONFLICT = 0).
// Epilogue: accumulators go through LDS so that every global store / residual load covers whole
// rows (16 B per lane, 1 KiB per wave-instruction).
#include "asrk_common.h"
#include "asrk_gemm.h"
#include "asrk_prof.h"

#include <cstdlib>

#ifdef ASRK_STAMPS
// Diagnostic build only (tools/build_diag.sh): per-workgroup s_memtime stamps, never in the product.
__device__ unsigned long long g_stamps[16384 * 8];
__device__ unsigned long long g_wall[16384 * 2];  // s_memrealtime (100 MHz) at stamps 0 and 4
extern "C" int asrk_debug_read_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#define STAMP(i) do { if (tid == 0 && blockIdx.x < 16384) { g_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    if ((i) == 0 || (i) == 4) g_wall[blockIdx.x * 2 + ((i) >> 2)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
extern "C" int asrk_debug_read_wall(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wall), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#else
#define STAMP(i) do { } while (0)
#endif

namespace {

constexpr int BM = 256;

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// Epilogue through LDS (see header): acc[nt][mt] of wave (wm, wn) -> out.  Every wave of the workgroup
// must have finished reading the K ring before it enters.
template <int WN, bool GELU, bool OUT_BF16>
__device__ __forceinline__ void epilogue256(const GemmArgs& a, f32x4 (&acc)[4][8], char* smem, int m0, int n0) {
    constexpr int NW = 2 * WN, BN = 64 * WN;
    constexpr int EP_ROW_BF16 = BN * 2 + 16;  // epilogue staging row strides (bytes)
    constexpr int EP_ROW_F32 = BN * 4 + 16;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15;
    const int fq = lane >> 4;
    STAMP(3);
    f32x4 bv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wn * 64 + nt * 16 + fq * 4;
        bv[nt] = (a.bias && n < a.N) ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (OUT_BF16) {
        // whole 256 x 256 bf16 tile, rows padded to 528 B (16-byte aligned, 2-way write conflicts at most)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int ml = wm * 128 + mt * 16 + frow;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int nl = wn * 64 + nt * 16 + fq * 4;
                const f32x4 v = gemm_epilogue_value<GELU, OUT_BF16>(a, acc[nt][mt], bv[nt], m0 + ml, n0 + nl);
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16_t)v[j];
                *reinterpret_cast<bf16x4*>(smem + ml * EP_ROW_BF16 + nl * 2) = o;
            }
        }
        STAMP(6);
        __syncthreads();
        STAMP(7);
        constexpr int LPR = BN / 8, RPI = 64 / LPR, ROWS_W = 256 / NW;  // lanes per row, rows per instruction, rows per wave
        const int rsub = lane / LPR, cch = lane % LPR;
        if (n0 + cch * 8 < a.N) {
            const int mfirst = m0 + wave * ROWS_W + rsub;
            int oq = mfirst / a.o_rpb, orem = mfirst % a.o_rpb;  // row map, stepped incrementally
#pragma unroll 4
            for (int i = 0; i < ROWS_W / RPI; ++i) {
                const int ml = wave * ROWS_W + i * RPI + rsub;
                if (m0 + ml < a.M) {
                    const int64_t orow = (int64_t)oq * a.o_bs + (int64_t)orem * a.o_rs + a.o_off;
                    const uint4 v = *reinterpret_cast<const uint4*>(smem + ml * EP_ROW_BF16 + cch * 16);
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.out) + orow + n0 + cch * 8) = v;
                }
                orem += RPI;
                while (orem >= a.o_rpb) { orem -= a.o_rpb; ++oq; }
            }
        }
    } else {
        // fp32 output (+ fp32 residual): two passes of 128 rows, rows padded to 1040 B.  The residual rows a
        // lane will add are fetched into registers ahead of time (pass 0: before the accumulators are
        // staged; pass 1: row by row as pass 0 consumes its copies), so their HBM latency runs under the
        // LDS staging and the barriers instead of in front of every store.
        constexpr int LPR = BN / 4, RPI = 64 / LPR, ROWS_W = 128 / NW, NIT = ROWS_W / RPI;
        const int rsub = lane / LPR, cch = lane % LPR;
        const bool col_ok = n0 + cch * 4 < a.N;
        f32x4 rv[NIT];
        int rrem_next = 0;  // residual row of (pass, i) being prefetched, stepped incrementally
        auto res_row = [&](int half, int i) -> f32x4 {
            // rows are visited in order (half 0: i = 0..NIT-1, then half 1); rrem_next is set at i == 0
            const int m = m0 + half * 128 + wave * ROWS_W + rsub + i * RPI;
            if (i == 0) rrem_next = m % a.res_rows;
            f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
            if (col_ok && m < a.M) r = *reinterpret_cast<const f32x4*>(a.res + (int64_t)rrem_next * a.ldr + n0 + cch * 4);
            rrem_next += RPI;
            while (rrem_next >= a.res_rows) rrem_next -= a.res_rows;
            return r;
        };
        if (a.res) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) rv[i] = res_row(0, i);
        } else {
#pragma unroll
            for (int i = 0; i < NIT; ++i) rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();
            if (wm == half) {
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const int ml = mt * 16 + frow;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int nl = wn * 64 + nt * 16 + fq * 4;
                        const f32x4 v = gemm_epilogue_value<GELU, OUT_BF16>(a, acc[nt][mt], bv[nt], m0 + half * 128 + ml, n0 + nl);
                        *reinterpret_cast<f32x4*>(smem + ml * EP_ROW_F32 + nl * 4) = v;
                    }
                }
            }
            __syncthreads();
            if (half == 0) STAMP(6);
            const int mfirst = m0 + half * 128 + wave * ROWS_W + rsub;
            int oq = mfirst / a.o_rpb, orem = mfirst % a.o_rpb;
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int ml = wave * ROWS_W + i * RPI + rsub;
                if (col_ok && mfirst + i * RPI < a.M) {
                    const int64_t orow = (int64_t)oq * a.o_bs + (int64_t)orem * a.o_rs + a.o_off;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(smem + ml * EP_ROW_F32 + cch * 16) + rv[i];
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + orow + n0 + cch * 4) = v;
                }
                if (half == 0 && a.res) rv[i] = res_row(1, i);
                orem += RPI;
                while (orem >= a.o_rpb) { orem -= a.o_rpb; ++oq; }
            }
            if (half == 0) STAMP(7);
        }
    }
    STAMP(4);
#ifdef ASRK_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(5);
#endif
}

// WN = waves along n (4: 256 x 256 tile, 512 threads, 1 workgroup / CU;  2: 256 x 128 tile, 256 threads,
// 2 workgroups / CU so that one's epilogue overlaps the other's K loop).
template <int WN, int BK, int NSTAGE, bool GELU, bool OUT_BF16>
__global__ __launch_bounds__(128 * WN, 2) void gemm256_kernel(const GemmArgs a) {
    constexpr int NTHR = 128 * WN, BN = 64 * WN;
    constexpr int RB = BK * 2;              // bytes per LDS row
    constexpr int CPR = RB / 16;            // 16-byte chunks per row (8 or 4)
    constexpr int XOPER = 256 * RB;         // bytes of the X slice per stage
    constexpr int STAGE = XOPER + BN * RB;
    constexpr int GX = 256 * CPR / NTHR, GW = BN * CPR / NTHR;  // global_load_lds per thread per stage
    constexpr int G = GX + GW;
    constexpr int RPR = NTHR / CPR;         // rows staged per round of NTHR lanes
    constexpr int KS = BK / 32;             // MFMA k-steps per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [NSTAGE][X slice | W slice]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- tile -> workgroup map.  Blocks b and b+8 share an XCD (and its 4 MiB L2).  The 8 XCDs
    //      form an (8/nsplit) x nsplit grid over (m, n): an XCD owns a contiguous run of m panels
    //      and 1/nsplit of the n tiles, walked n-fastest.
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int ns = a.nsplit, mgc = 8 / ns;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tn_per = tiles_n / ns, tm_per = (tiles_m + mgc - 1) / mgc;
    const int tm = (xcd / ns) * tm_per + idx / tn_per;
    if (idx / tn_per >= tm_per || tm >= tiles_m) return;  // padding blocks of the XCD grid
    const int m0 = tm * BM;
    const int n0 = ((xcd % ns) * tn_per + idx % tn_per) * BN;

    // ---- per-thread staging sources (CPR/2 rows of X and of W per stage)
    const int srow = tid / CPR;  // + RPR * i   (RPR is a multiple of 16, so the swizzle term is the same for all i)
    const int sswz = CPR == 8 ? (srow & 7) : (((srow >> 3) & 1) << 1);
    const int chunk = (tid % CPR) ^ sswz;
    const bf16_t* xp[GX];
    const bf16_t* wp[GW];
#pragma unroll
    for (int i = 0; i < GX; ++i) {
        int m = m0 + srow + RPR * i;
        m = m < a.M ? m : a.M - 1;
        xp[i] = a.X + (int64_t)(m / a.x_rpb) * a.x_bs + (int64_t)(m % a.x_rpb) * a.x_rs + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < GW; ++i) {
        int n = n0 + srow + RPR * i;
        n = n < a.N ? n : a.N - 1;
        wp[i] = a.W + (int64_t)n * a.K + chunk * 8;
    }
    const int wave_lds = wave * 1024;

    auto stage = [&](int kt, int slot) {
        char* sb = smem + slot * STAGE;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < GX; ++i) glds16(xp[i] + k0, sb + i * (NTHR * 16) + wave_lds);
#pragma unroll
        for (int i = 0; i < GW; ++i) glds16(wp[i] + k0, sb + XOPER + i * (NTHR * 16) + wave_lds);
    };

    f32x4 acc[4][8];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fq = lane >> 4;
    const int fswz = CPR == 8 ? (frow & 7) : (((frow >> 3) & 1) << 1);
    const int x_off = (wm * 128 + frow) * RB;
    const int w_off = XOPER + (wn * 64 + frow) * RB;

    auto compute = [&](const char* sb) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int coff = ((ks * 4 + fq) ^ fswz) << 4;
            bf16x8 wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(sb + w_off + t * 16 * RB + coff);
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                bf16x8 xf[4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    xf[t] = *reinterpret_cast<const bf16x8*>(sb + x_off + (mh * 4 + t) * 16 * RB + coff);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[nt][mh * 4 + t] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[t], acc[nt][mh * 4 + t], 0, 0, 0);
            }
        }
    };

    // ---- start-up skew (first round of workgroups only): all tiles of a GEMM take the same time, so
    //      without it every CU reaches its epilogue at the same moment and the output stores arrive as
    //      one HBM write burst per round; spreading the phases keeps the stores under the other CUs' K loops.
    if (a.skew > 0 && blockIdx.x < 256) {
        const int n = ((blockIdx.x >> 3) & 7) * a.skew;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int nkt = a.K / BK;
    STAMP(0);
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nkt) stage(s, s);
    int slot = 0, fill = NSTAGE - 1;  // ring slot of slice t / of slice t + NSTAGE - 1
    for (int kt = 0; kt < nkt; ++kt) {
        const int ahead = nkt - 1 - kt;  // slices after this one whose loads are already issued (capped below)
        if (ahead >= NSTAGE - 2) wait_vm_barrier<(NSTAGE - 2) * G>();
        else if (NSTAGE > 3 && ahead == 2) wait_vm_barrier<2 * G>();
        else if (NSTAGE > 2 && ahead == 1) wait_vm_barrier<1 * G>();
        else wait_vm_barrier<0>();
        if (kt == 1) STAMP(2);
        if (kt + NSTAGE - 1 < nkt) stage(kt + NSTAGE - 1, fill);
        if (kt == 0) STAMP(1);
        compute(smem + slot * STAGE);
        slot = slot + 1 == NSTAGE ? 0 : slot + 1;
        fill = fill + 1 == NSTAGE ? 0 : fill + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone is done with the ring

    epilogue256<WN, GELU, OUT_BF16>(a, acc, smem, m0, n0);
}

template <int WN, int BK, int NSTAGE, bool GELU, bool OUT_BF16>
int launch(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    constexpr int BN = 64 * WN;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    int ns = 1;
    a.skew = 0;
#ifdef ASRK_DIAG  // diagnostic build only: XCD n-splits and start-up skew (both measured, neither kept: DESIGN.md 4.2)
    static const int forced = [] { const char* e = getenv("ASRK_GEMM_NSPLIT"); return e ? atoi(e) : 0; }();
    if (forced > 0 && (forced == 1 || forced == 2 || forced == 4 || forced == 8) && tiles_n % forced == 0) ns = forced;
    static const int skew = [] { const char* e = getenv("ASRK_GEMM_SKEW"); return e ? atoi(e) : 0; }();
    a.skew = skew;
#endif
    a.nsplit = ns;
    const int tm_per = (tiles_m + 8 / ns - 1) / (8 / ns);
    const int tiles = 8 * tm_per * (tiles_n / ns);
    constexpr size_t ring = (size_t)NSTAGE * (256 + BN) * BK * 2;
    constexpr size_t ep = (size_t)256 * (BN * 2 + 16) > (size_t)128 * (BN * 4 + 16) ? (size_t)256 * (BN * 2 + 16)
                                                                                     : (size_t)128 * (BN * 4 + 16);
    constexpr size_t lds = ring > ep ? ring : ep;
    static_assert(lds <= (WN == 4 ? 163840 : 81920), "LDS budget");
    auto kern = gemm256_kernel<WN, BK, NSTAGE, GELU, OUT_BF16>;
    static std::atomic<uint64_t> lds_set{0};
    if (asrk_max_lds_once((const void*)kern, (int)lds, lds_set) != ASRK_OK) return ASRK_ERR_HIP;
    AsrkProfScope prof(ASRK_PROF_GEMM256, 2.0 * a.M * a.N * a.K, st);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(128 * WN), lds, st, a);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}

template <int WN, int BK, int NSTAGE>
int dispatch(const GemmArgs& a, bool gelu, bool out_bf16, hipStream_t st) {
    if (gelu) return out_bf16 ? launch<WN, BK, NSTAGE, true, true>(a, st) : launch<WN, BK, NSTAGE, true, false>(a, st);
    return out_bf16 ? launch<WN, BK, NSTAGE, false, true>(a, st) : launch<WN, BK, NSTAGE, false, false>(a, st);
}

}  // namespace

// Arguments are validated by asrk_gemm_launch (asrk_gemm.hip), which dispatches here.
int asrk_gemm256_launch(const GemmArgs& a, bool gelu, bool out_bf16, hipStream_t st) {
    if (a.splits != 1 || (a.N % 8) != 0) return ASRK_ERR_SHAPE;
    // Default: bf16 outputs go to the persistent kernel (asrk_gemm256p.hip), fp32 outputs (+ residual) to the
    // one-tile-per-workgroup kernel below, whose two-pass epilogue has room for a deeper residual prefetch.
#ifdef ASRK_DIAG
    // diagnostic build only, ASRK_GEMM_CFG: 2 = one tile per workgroup always, 4 = persistent always,
    // 1 = 256x256 BK 32 x 4 stages, 3 = 256x128 BK 32 x 3 stages (variants measured and not kept: DESIGN.md 4.2)
    static const int cfg = [] { const char* e = getenv("ASRK_GEMM_CFG"); return e ? atoi(e) : 0; }();
    if (cfg == 1) return dispatch<4, 32, 4>(a, gelu, out_bf16, st);
    if (cfg == 3) return dispatch<2, 32, 3>(a, gelu, out_bf16, st);
#else
    constexpr int cfg = 0;
#endif
    if ((cfg == 4 || (cfg == 0 && out_bf16)) && a.K % 64 == 0 && a.K >= 192 && !(gelu && a.dg_pre) && asrk_gemm256p_fits(a))
        return asrk_gemm256p_launch(a, gelu, out_bf16, st);
    return dispatch<4, 64, 2>(a, gelu, out_bf16, st);
}
