// Persistent form of the 256 x 256-tile bf16 MFMA GEMM (contract of asrk_gemm.hip, tile and K loop of
// asrk_gemm256.hip): one 512-thread workgroup per CU walks a list of tiles, and the K stream never stops
// at a tile boundary.
//
//   * The LDS ring (2 stages x 64 K-columns x (256 + 256) rows = 128 KiB) is treated as one long stream of
//     K slices: while the last slice of tile i is multiplied, slice 0 of tile i+1 is already being written
//     into the other slot by global_load_lds, so a tile does not start with an exposed HBM/L2 round trip
//     (5-8 k cycles of ~56 k per tile in the one-tile-per-workgroup kernel).
//   * The epilogue therefore cannot borrow the ring.  Each wave transposes its own 128 x 64 block through a
//     private 4 KiB LDS patch (8 waves x 4 KiB = the 32 KiB the ring leaves free), 16 rows at a time:
//     accumulator layout in, whole 128-byte (bf16) / 256-byte (fp32) row segments out, 16 B per lane.  A wave's
//     LDS operations execute in order, so this needs no workgroup barrier at all: the waves drift apart in
//     the epilogue and meet again at the first barrier of the next tile.
//   * fp32 residual rows are fetched one 16-row group ahead of the group being stored.
//
// Tile order: workgroup b runs on XCD b & 7 (verified with HW_REG_XCC_ID); XCD x owns a contiguous run of
// m panels and walks its tiles n-fastest, 32 tiles (one per CU of the XCD) at a time, so the 32 concurrent
// tiles of an XCD share a few X panels and the whole weight matrix in that XCD's L2.
#include "asrk_common.h"
#include "asrk_gemm.h"
#include "asrk_prof.h"

#include <cstdlib>

#ifdef ASRK_STAMPS
__device__ unsigned long long g_pstamps[1024 * 8];
extern "C" int asrk_debug_read_pstamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pstamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#define PSTAMP(i) do { if (tid == 0 && blockIdx.x < 1024) g_pstamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif

namespace {

// MT = 16-row accumulator blocks per wave: 8 -> 256-row tiles, 6 -> 192-row tiles.  The tile height is a launch-time
// choice because the tile count rarely divides by the CU count (schedule_cost below); at the encoder's shapes the
// 256-row tile wins everywhere, the 192-row tile only pays when the last round of 256-row tiles is nearly empty.
constexpr int BN = 256, BK = 64, NTHR = 512;
constexpr int RB = BK * 2;           // bytes per LDS row
constexpr int CPR = RB / 16;         // 16-byte chunks per row
constexpr int GW = BN * CPR / NTHR;
constexpr int RPR = NTHR / CPR;      // rows staged per round of NTHR lanes
constexpr int STAGE_MAX = (256 + BN) * RB;
constexpr int RING = 2 * STAGE_MAX;  // the ring is sized for the 256-row tile
constexpr int PATCH = 4096;          // private epilogue patch per wave
constexpr int LDS_BYTES = RING + 8 * PATCH;
static_assert(LDS_BYTES == 163840, "uses all of the CU's LDS");

__device__ __forceinline__ void wait_all_barrier() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }
// waits for everything but this wave's N youngest vector-memory operations
template <int N>
__device__ __forceinline__ void wait_older() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Orders this wave's LDS stores before its following LDS loads of other lanes' data (the hardware runs a
// wave's LDS operations in order; this only stops the compiler from moving them).
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS accesses of the fast epilogue, as asm: the compiler puts s_waitcnt vmcnt(0) in front of any LDS access of its own
// that follows a global_load_lds it cannot tell apart (here: slice 1 of the next tile, issued just before the
// epilogue and wanted two slices later), which would expose a full memory round trip per tile.  LDS operations of a wave
// complete in order; the waits are explicit.
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return lds_addr32(p); }
template <int OFF>
__device__ __forceinline__ void lds_write8(uint32_t addr, u32x2 v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ u32x4 lds_read16(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// waits until at most N LDS operations of this wave are outstanding; the operands tie the loaded values to the wait
template <int N>
__device__ __forceinline__ void lds_wait(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_DGELU = 2 };  // DGELU: times gelu'(dg_pre[m][n]) (projector backward)

// One wave's 128 x 64 block (rows mw.., columns nw..) from accumulators to memory.  The caller has put this block's 64
// bias values at the start of the patch (one float per lane).
template <int MT, int EPI, bool OUT_BF16>
__device__ __forceinline__ void epilogue_wave(const GemmArgs& a, f32x4 (&acc)[4][MT], char* patch, int mw, int nw, int lane) {
    constexpr bool GELU = EPI == EPI_GELU;
    const int frow = lane & 15, fq = lane >> 4;
    // a lane needs the bias of columns nw + 16 nt + 4 fq .. +3
    f32x4 bv[4];
    wave_lds_order();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bv[nt] = *reinterpret_cast<const f32x4*>(patch + (nt * 16 + fq * 4) * 4);
    wave_lds_order();
    const float alpha = a.alpha;  // locals, not a.*: see gemm_epilogue_value
    const bf16_t* const dg_pre = EPI == EPI_DGELU ? a.dg_pre : nullptr;
    const int64_t ld_dg = a.ld_dg;
    const int M = a.M, N = a.N;
    if (OUT_BF16 && EPI != EPI_DGELU) {
        // Fast path (every tile of the encoder's GEMMs but the ragged edge): the block lies inside the matrix and
        // inside one block of the output row map, so row r of it is at base + r * o_rs with a wave-uniform base.
        // Straight-line code: no per-row division, no 64-bit lane arithmetic, no predication -- the general path
        // below spends 8.5 k cycles per tile on exactly those, stores or no stores.
        const int q0 = mw / a.o_rpb, r0 = mw - q0 * a.o_rpb;
        if (mw + 16 * MT <= M && nw + 64 <= N && r0 + 16 * MT <= a.o_rpb && a.o_rs < (1 << 24)) {
            const int r8 = lane >> 3, c8 = lane & 7;
            const int o_rs = (int)a.o_rs;
            bf16_t* const base = reinterpret_cast<bf16_t*>(a.out) + a.o_off + (int64_t)q0 * a.o_bs + (int64_t)r0 * a.o_rs + nw;
            const int lane_off = r8 * o_rs + c8 * 8;
            // two [16][64] bf16 half-patches used alternately (layout: see the general path below)
            uint32_t wa[4], ra[2];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wa[nt] = lds_addr(patch) + frow * 128 + (((nt * 4 + fq) ^ frow) << 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) ra[h] = lds_addr(patch) + (h * 8 + r8) * 128 + ((c8 ^ ((h * 8 + r8) >> 1)) << 4);
            auto put = [&](auto mt_c) {  // 16 rows of accumulators -> half-patch mt & 1
                constexpr int mt = decltype(mt_c)::value;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    f32x4 v = acc[nt][mt] + bv[nt];
                    if (GELU) {
                        const f32x2_t lo = gelu_erf_x2(f32x2_t{v[0], v[1]}), hi = gelu_erf_x2(f32x2_t{v[2], v[3]});
                        v = f32x4{lo.x, lo.y, hi.x, hi.y} * (0.5f * alpha);
                    } else {
                        v *= alpha;
                    }
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (bf16_t)v[j];
                    lds_write8<(mt & 1) * 2048>(wa[nt], __builtin_bit_cast(u32x2, o));
                }
            };
            put(std::integral_constant<int, 0>{});
            static_for<MT>([&](auto mt_c) {
                constexpr int mt = decltype(mt_c)::value;
                u32x4 v0 = lds_read16<(mt & 1) * 2048>(ra[0]);
                u32x4 v1 = lds_read16<(mt & 1) * 2048>(ra[1]);
                if constexpr (mt + 1 < MT) {
                    put(std::integral_constant<int, mt + 1>{});
                    lds_wait<4>(v0, v1);  // in order: only the 4 writes of the next group may still be outstanding
                } else {
                    lds_wait<0>(v0, v1);
                }
                if (r8 & 1) {  // rows h * 8 + r8: odd rows hold their 8-byte halves swapped
                    v0 = u32x4{v0.z, v0.w, v0.x, v0.y};
                    v1 = u32x4{v1.z, v1.w, v1.x, v1.y};
                }
                // non-temporal: the tile is not read again by this kernel, so it should not displace the X panels and W
                // from the caches (pipeline +0.5 %, and better than keeping the 147 MB outputs cacheable for the LayerNorm that follows; a GEMM repeated on the same operands +4-12 %, its X then stays in
                // the 256 MB memory-side cache between launches)
                __builtin_nontemporal_store(v0, reinterpret_cast<u32x4*>(base + (int64_t)((mt * 16) * o_rs) + lane_off));
                __builtin_nontemporal_store(v1, reinterpret_cast<u32x4*>(base + (int64_t)((mt * 16 + 8) * o_rs) + lane_off));
            });
            return;
        }
    }
    if (OUT_BF16) {
        // Two [16][64] bf16 half-patches (2 KiB each), used alternately so the stores of one 16-row group overlap
        // the staging of the next.  8-byte unit u of row r lives at unit u ^ r: conflict-free for the
        // accumulator-layout ds_write_b64 (16 rows x one unit) and for the row-layout ds_read_b128 (a 16-byte
        // chunk is the unit pair 2c, 2c+1 -> pair c ^ (r >> 1), halves swapped when r is odd).
        const int r8 = lane >> 3, c8 = lane & 7;
        const bool col_ok = nw + c8 * 8 < a.N;
        int oq = (mw + r8) / a.o_rpb, orem = (mw + r8) % a.o_rpb;  // row map of the rows this lane stores, stepped by 8
        bf16_t* const out = reinterpret_cast<bf16_t*>(a.out) + a.o_off + nw + c8 * 8;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            char* half = patch + (mt & 1) * 2048;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int m = mw + mt * 16 + frow, n = nw + nt * 16 + fq * 4;
                const f32x4 v = gemm_epilogue_value<GELU>(alpha, dg_pre, ld_dg, m < M && n < N, acc[nt][mt], bv[nt], m, n);
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16_t)v[j];
                *reinterpret_cast<bf16x4*>(half + frow * 128 + (((nt * 4 + fq) ^ frow) << 3)) = o;
            }
            wave_lds_order();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = h * 8 + r8;
                uint4 v = *reinterpret_cast<const uint4*>(half + row * 128 + ((c8 ^ (row >> 1)) << 4));
                if (row & 1) v = uint4{v.z, v.w, v.x, v.y};
#ifdef ASRK_NOSTORE  // diagnostic: epilogue cost without its global stores (a never-true condition keeps the data path)
                if (col_ok && mw + mt * 16 + row < a.M && v.x == 0x7fc07fc1u && v.y == 0x12345678u)
#else
                if (col_ok && mw + mt * 16 + row < a.M)
#endif
                    *reinterpret_cast<uint4*>(out + (int64_t)oq * a.o_bs + (int64_t)orem * a.o_rs) = v;
                orem += 8;
                while (orem >= a.o_rpb) { orem -= a.o_rpb; ++oq; }
            }
        }
        wave_lds_order();
    } else {
        // [16][64] fp32 patch, exactly 4 KiB: 16-byte chunk c of row r lives at chunk c ^ r (conflict-free for
        // the accumulator-layout ds_write_b128 and for the row-layout ds_read_b128)
        const int r4 = lane >> 4, c16 = lane & 15;
        const bool col_ok = nw + c16 * 4 < a.N;
        int oq = (mw + r4) / a.o_rpb, orem = (mw + r4) % a.o_rpb;  // rows this lane stores, stepped by 4
        float* const out = reinterpret_cast<float*>(a.out) + a.o_off + nw + c16 * 4;
        // residual rows, fetched one 16-row group ahead
        int rrem = a.res ? (mw + r4) % a.res_rows : 0;
        const float* const res = a.res ? a.res + nw + c16 * 4 : nullptr;
        f32x4 rv[2][4];
        auto fetch = [&](int mt, f32x4 (&dst)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dst[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (res && col_ok && mw + mt * 16 + i * 4 + r4 < a.M)
                    dst[i] = *reinterpret_cast<const f32x4*>(res + (int64_t)rrem * a.ldr);
                rrem += 4;
                while (rrem >= a.res_rows) rrem -= a.res_rows;
            }
        };
        fetch(0, rv[0]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt + 1 < MT) fetch(mt + 1, rv[(mt + 1) & 1]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int m = mw + mt * 16 + frow, n = nw + nt * 16 + fq * 4;
                const f32x4 v = gemm_epilogue_value<GELU>(alpha, dg_pre, ld_dg, m < M && n < N, acc[nt][mt], bv[nt], m, n);
                *reinterpret_cast<f32x4*>(patch + frow * 256 + (((nt * 4 + fq) ^ frow) << 4)) = v;
            }
            wave_lds_order();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 4 + r4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(patch + row * 256 + ((c16 ^ row) << 4)) + rv[mt & 1][i];
                if (col_ok && mw + mt * 16 + row < a.M)
                    *reinterpret_cast<f32x4*>(out + (int64_t)oq * a.o_bs + (int64_t)orem * a.o_rs) = v;
                orem += 4;
                while (orem >= a.o_rpb) { orem -= a.o_rpb; ++oq; }
            }
            wave_lds_order();
        }
    }
}

template <int MT, int EPI, bool OUT_BF16>
__global__ __launch_bounds__(NTHR, 2) void gemm256p_kernel(const GemmArgs a) {
    constexpr int BM = 32 * MT;          // 2 waves along m, 16 * MT rows each
    constexpr int XOPER = BM * RB;       // bytes of the X slice
    constexpr int STAGE = XOPER + BN * RB;
    constexpr int GX = BM * CPR / NTHR;
    static_assert(BM * CPR % NTHR == 0 && MT % 2 == 0, "tile height");
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][X slice | W slice] | 8 epilogue patches
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // ---- this workgroup's tile list
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int xcd = blockIdx.x & 7, lidx = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    const int tm_per = (tiles_m + 7) / 8;
    const int tm_lo = xcd * tm_per;
    const int tm_cnt = tiles_m - tm_lo < tm_per ? tiles_m - tm_lo : tm_per;
    const int count = tm_cnt > 0 ? tm_cnt * tiles_n : 0;
    if (lidx >= count) return;

    // ---- per-thread staging sources of the current tile (GX rows of X and GW rows of W per slice)
    const int srow = tid / CPR;
    const int chunk = (tid % CPR) ^ (srow & 7);
    const bf16_t* xp[GX];
    const bf16_t* wp[GW];
    auto set_sources = [&](int m0, int n0) {
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
    };
    const int wave_lds = wave * 1024;
    auto stage = [&](int kt, int slot) {
        char* sb = smem + slot * STAGE;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < GX; ++i) glds16(xp[i] + k0, sb + i * (NTHR * 16) + wave_lds);
#pragma unroll
        for (int i = 0; i < GW; ++i) glds16(wp[i] + k0, sb + XOPER + i * (NTHR * 16) + wave_lds);
    };

    f32x4 acc[4][MT];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const int x_off = (wm * (16 * MT) + frow) * RB;
    const int w_off = XOPER + (wn * 64 + frow) * RB;
    auto compute = [&](const char* sb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + fq) ^ (frow & 7)) << 4;
            bf16x8 wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(sb + w_off + t * 16 * RB + coff);
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                constexpr int HB = MT / 2;
                bf16x8 xf[HB];
#pragma unroll
                for (int t = 0; t < HB; ++t)
                    xf[t] = *reinterpret_cast<const bf16x8*>(sb + x_off + (mh * HB + t) * 16 * RB + coff);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int t = 0; t < HB; ++t)
                        acc[nt][mh * HB + t] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[t], acc[nt][mh * HB + t], 0, 0, 0);
            }
        }
    };

    const int nkt = a.K / BK;
    float bias_lane = 0.f;
    int t = lidx;
    int m0 = (tm_lo + t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
    set_sources(m0, n0);
    PSTAMP(0);
    stage(0, 0);
    stage(1, 1);
    wait_older<GX + GW>();  // my part of slice 0 ...
    asm volatile("s_barrier" ::: "memory");  // ... and everyone's
    int slot = 0;
#ifdef ASRK_STAMPS
    int tile_no = 0;
#endif
    // A tile enters its K loop with slices 0 and 1 staged (slice 1 possibly still in flight).  Slice kt + 1 is
    // staged at the top of iteration kt >= 1; the last iteration stages slice 0 of the next tile, and slice 1 of
    // the next tile follows right after the loop, BEFORE the epilogue: a wave's vector-memory operations complete in
    // issue order, so loads issued behind the epilogue's 16 stores wait until those have drained (measured:
    // 7.6 k cycles per tile lost in the K loop that followed the epilogue when slice 1 was staged after it).
    while (true) {
        const int tnext = t + wgs_per_xcd;
        const bool has_next = tnext < count;
        int m0n = 0, n0n = 0;
#pragma nounroll  // also keeps the kt == 0 case from being peeled into a second copy of the body (spills)
        for (int kt = 0; kt < nkt; ++kt) {
            // my loads of this slice have landed; after the barrier everyone's have, and everyone has finished
            // reading the other slot.  Slice 0 needs neither: it was complete for every wave at the barrier in front
            // of the previous epilogue (or of the loop), and nothing is staged at kt == 0 -- so a wave that leaves
            // its epilogue early starts on slice 0 while its SIMD partner is still storing.
            if (kt > 0) wait_all_barrier();
#ifdef ASRK_STAMPS
            if (kt == 0) { if (tile_no == 0) PSTAMP(1); else if (tile_no == 1) PSTAMP(5); }
#endif
            if (kt >= 1 && kt + 1 < nkt) stage(kt + 1, slot ^ 1);
            if (kt + 1 == nkt) {
                // this tile's bias, one column per lane (spread to the accumulator layout through the LDS patch in
                // the epilogue), issued before the next tile's loads so that it is older than all of them
                const int n = n0 + wn * 64 + lane;
                bias_lane = (a.bias && n < a.N) ? a.bias[n] : 0.f;
                if (has_next) {  // the K stream runs on into the next tile
                    m0n = (tm_lo + tnext / tiles_n) * BM;
                    n0n = (tnext % tiles_n) * BN;
                    set_sources(m0n, n0n);
                    stage(0, slot ^ 1);
                }
            }
            compute(smem + slot * STAGE);
            slot ^= 1;
        }
        // The bias goes to the patch first: the wait for its load (older than every load of the next tile) sits here,
        // where those have had a whole slice of MFMAs to land, and the epilogue proper then never waits on vmcnt.
        char* const patch = smem + RING + wave * PATCH;
        reinterpret_cast<float*>(patch)[lane] = bias_lane;
        if (has_next) {
            // slice 0 of the next tile has landed; once everyone has finished reading the slot of the last slice,
            // slice 1 of the next tile goes there
            wait_all_barrier();
            stage(1, slot ^ 1);
        }
#ifdef ASRK_STAMPS
        if (tile_no == 0) PSTAMP(2);
#endif
        epilogue_wave<MT, EPI, OUT_BF16>(a, acc, patch, m0 + wm * (16 * MT), n0 + wn * 64, lane);
#ifdef ASRK_STAMPS
        if (tile_no == 0) PSTAMP(3); else if (tile_no == 1) PSTAMP(6);
        ++tile_no;
#endif
        if (!has_next) break;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        t = tnext;
        m0 = m0n;
        n0 = n0n;
    }
    PSTAMP(7);
}

int device_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) return 256;
        return n;
    }();
    return cus;
}

// Time of the whole GEMM in units of "one 256-row tile on one CU": rounds of the fullest CU x relative tile cost.
double schedule_cost(const GemmArgs& a, int mt) {
    const int bm = 32 * mt;
    const long tiles_m = (a.M + bm - 1) / bm, tiles_n = (a.N + BN - 1) / BN;
    const long per_xcd = ((tiles_m + 7) / 8) * tiles_n;
    const long wgs = device_cus() / 8;
    const long rounds = (per_xcd + wgs - 1) / wgs;
    // a 192-row tile does 3/4 of the MFMAs but loads the same W slice and pays the same barriers: measured 0.84 of a
    // 256-row tile (fc2 at M = 96000: 6 rounds of 192 rows = 437 us against 5 rounds of 256 rows = 432 us)
    return (double)rounds * (mt == 8 ? 1.0 : 0.84);
}

template <int MT, int EPI, bool OUT_BF16>
int launch(const GemmArgs& a, hipStream_t st) {
    constexpr int BM = 32 * MT;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int per_xcd = ((tiles_m + 7) / 8) * tiles_n;  // tiles of the fullest XCD
    int wgs_per_xcd = device_cus() / 8;
    if (per_xcd < wgs_per_xcd) wgs_per_xcd = per_xcd;
    auto kern = gemm256p_kernel<MT, EPI, OUT_BF16>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
        return ASRK_ERR_HIP;
    AsrkProfScope prof(ASRK_PROF_GEMM, 2.0 * a.M * a.N * a.K, st);
    hipLaunchKernelGGL(kern, dim3(8 * wgs_per_xcd), dim3(NTHR), LDS_BYTES, st, a);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}

template <int MT>
int dispatch(const GemmArgs& a, bool gelu, bool out_bf16, hipStream_t st) {
    if (a.dg_pre) {
        if (gelu) return ASRK_ERR_SHAPE;  // gelu and gelu' never combine (asrk_gemm.hip's contract)
        return out_bf16 ? launch<MT, EPI_DGELU, true>(a, st) : launch<MT, EPI_DGELU, false>(a, st);
    }
    if (gelu) return out_bf16 ? launch<MT, EPI_GELU, true>(a, st) : launch<MT, EPI_GELU, false>(a, st);
    return out_bf16 ? launch<MT, EPI_NONE, true>(a, st) : launch<MT, EPI_NONE, false>(a, st);
}

}  // namespace

// Arguments are validated by asrk_gemm_launch (asrk_gemm.hip); needs splits == 1, N % 8 == 0, K % 64 == 0.
int asrk_gemm256p_launch(const GemmArgs& a, bool gelu, bool out_bf16, hipStream_t st) {
    if (a.splits != 1 || (a.N % 8) != 0 || (a.K % BK) != 0 || a.K < 2 * BK) return ASRK_ERR_SHAPE;
    // ASRK_GEMM_MT (A/B measurements only): 8 or 6 forces the tile height
    static const int forced = [] { const char* e = getenv("ASRK_GEMM_MT"); return e ? atoi(e) : 0; }();
    const bool use6 = forced == 6 || (forced != 8 && schedule_cost(a, 6) < schedule_cost(a, 8));
    return use6 ? dispatch<6>(a, gelu, out_bf16, st) : dispatch<8>(a, gelu, out_bf16, st);
}
