// Persistent form of the 256 x 256-tile bf16 MFMA GEMM (contract of asrk_gemm.hip, tile and K loop of
// asrk_gemm256.hip): one 512-thread workgroup per CU walks a list of tiles, and the K stream never stops
// at a tile boundary.
//
//   * The LDS (160 KiB) is a ring of K slices that never stops at a tile boundary: three 32 KiB slots for X slices and
//     two for W slices, filled by global_load_lds.  Phase D of slice j stages W of slice j + 2 and X of slice j + 3 --
//     slices that near the end of a tile belong to the next tile, so a tile does not start with an exposed memory
//     round trip and its first loads are issued before the previous epilogue's stores (a wave's vector-memory
//     operations complete in issue order).
//   * The epilogue cannot borrow the live ring slots.  Each wave transposes its own 128 x 64 block through a private
//     4 KiB patch, 16 rows at a time: accumulator layout in, whole 128-byte (bf16) / 256-byte (fp32) row segments out,
//     16 B per lane.  The 8 patches alias the X slot the tile's last slice has just left (the X batch that would go
//     there is staged one slice later).  A wave's LDS operations execute in order, so this needs no workgroup barrier:
//     the waves drift apart in the epilogue and meet again at the first barrier of the next tile.
//   * The K loop is hand-scheduled: fragments by ds_read_b128 asm one phase ahead with counted lgkmcnt waits, the DMA
//     batch spread behind the MFMA groups of the barrier phase (see the kernel).
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
constexpr int SLOT = 256 * RB;        // one operand slice of up to 256 rows: 32 KiB
constexpr int NXS = 3, NWS = 2;      // ring slots of X and of W
constexpr int WBASE = NXS * SLOT;    // W slots follow the X slots
constexpr int PATCH = 4096;          // private epilogue patch per wave; the 8 patches alias the free X slot
constexpr int LDS_BYTES = (NXS + NWS) * SLOT;
static_assert(8 * PATCH == SLOT, "the epilogue patches fill exactly one X slot");
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
    {
        const uint32_t ba = lds_addr(patch) + fq * 16;
        u32x4 b0 = lds_read16<0>(ba), b1 = lds_read16<64>(ba), b2 = lds_read16<128>(ba), b3 = lds_read16<192>(ba);
        lds_wait<0>(b0, b1);
        lds_wait<0>(b2, b3);
        bv[0] = __builtin_bit_cast(f32x4, b0);
        bv[1] = __builtin_bit_cast(f32x4, b1);
        bv[2] = __builtin_bit_cast(f32x4, b2);
        bv[3] = __builtin_bit_cast(f32x4, b3);
    }
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
                        const f32x2_t lo = gelu2_x2(f32x2_t{v[0], v[1]}), hi = gelu2_x2(f32x2_t{v[2], v[3]});
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
                const f32x4 v = gemm_epilogue_value<GELU, true>(alpha, dg_pre, ld_dg, m < M && n < N, acc[nt][mt], bv[nt], m, n);
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
                const f32x4 v = gemm_epilogue_value<GELU, false>(alpha, dg_pre, ld_dg, m < M && n < N, acc[nt][mt], bv[nt], m, n);
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
    constexpr int BM = 32 * MT;  // 2 waves along m, 16 * MT rows each
    constexpr int GX = BM * CPR / NTHR;
    static_assert(BM * CPR % NTHR == 0 && MT % 2 == 0, "tile height");
    // LDS: X slots 0..2 | W slots 0..1.  Slice j of the K stream lives in X slot j % 3 and W slot j % 2.
    extern __shared__ __attribute__((aligned(16))) char smem[];
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

    // ---- per-thread DMA sources (GX rows of X and GW rows of W per slice): 32-bit byte offsets from the
    // (wave-uniform) tensor bases, so the DMA takes the base in SGPRs and one offset VGPR per lane instead of a 64-bit
    // address pair (the launcher checks that both tensors span < 4 GiB).  X and W switch to the next tile at different
    // points of the stream, hence two setters.
    const int srow = tid / CPR;
    const int chunk = (tid % CPR) ^ (srow & 7);
    uint32_t xo[GX], wo[GW];
    auto set_x = [&](int m0) {
#pragma unroll
        for (int i = 0; i < GX; ++i) {
            int m = m0 + srow + RPR * i;
            m = m < a.M ? m : a.M - 1;
            xo[i] = (uint32_t)(((int64_t)(m / a.x_rpb) * a.x_bs + (int64_t)(m % a.x_rpb) * a.x_rs + chunk * 8) * 2);
        }
    };
    auto set_w = [&](int n0) {
#pragma unroll
        for (int i = 0; i < GW; ++i) {
            int n = n0 + srow + RPR * i;
            n = n < a.N ? n : a.N - 1;
            wo[i] = (uint32_t)(((int64_t)n * a.K + chunk * 8) * 2);
        }
    };
    const int wave_lds = wave * 1024;
    // half h of the X (W) DMAs of slice kt into X slot xs (W slot ws)
#ifdef ASRK_ABLATE  // diagnostic builds only (tools/gemm_ablate.sh), after the prologue: 1 = no DMA, 2 = no
    // fragment reads, 4 = every X DMA re-reads slice 0 of the first tile (cache hits), 8 = the same for W
    bool ablate_on = false;
#define ASRK_ABLATED(what) ((ASRK_ABLATE & (what)) && ablate_on)
#else
#define ASRK_ABLATED(what) false
#endif
    auto stage_x = [&](int kt, int xs, auto hc) {
        constexpr int h = decltype(hc)::value;
        if (ASRK_ABLATED(1)) return;
        if (ASRK_ABLATED(4)) kt = 0;
        char* sb = smem + xs * SLOT;
        const char* xb = uniform_ptr(reinterpret_cast<const char*>(a.X) + (size_t)kt * (BK * 2));
#pragma unroll
        for (int i = h; i < GX; i += 2) glds16(xb + xo[i], sb + i * (NTHR * 16) + wave_lds);
    };
    auto stage_w = [&](int kt, int ws, auto hc) {
        constexpr int h = decltype(hc)::value;
        if (ASRK_ABLATED(1)) return;
        if (ASRK_ABLATED(8)) kt = 0;
        char* sb = smem + WBASE + ws * SLOT;
        const char* wb = uniform_ptr(reinterpret_cast<const char*>(a.W) + (size_t)kt * (BK * 2));
#pragma unroll
        for (int i = h; i < GW; i += 2) glds16(wb + wo[i], sb + i * (NTHR * 16) + wave_lds);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    f32x4 acc[4][MT];  // [nt][mt]; first written by the MFMAs of a tile's first k-step (see group)

    const int frow = lane & 15, fq = lane >> 4;
    constexpr int HB = MT / 2;  // 16-row X fragments per half of the wave's rows

    // ---- fragment stream.  A slice is multiplied in four phases (k-step ks, row half mh) of 4 x HB MFMAs:
    //   A = (0, 0), B = (0, 1), C = (1, 0), D = (1, 1).  Fragments live in W[4] (one per 16 columns), X[HB] (even
    // phases) and Xn[HB] (odd phases) and are read one phase ahead of their use, by hand-written ds_read_b128 with
    // counted lgkmcnt waits (LDS returns in order), each register set reloaded as soon as its last MFMA has issued:
    //   even phase: X and W[0] ready at its start (lgkmcnt(3): W[1..3] may still be on their way), W[g] before group
    //               g; reads Xn (same slot, same k-step) behind group 0;
    //   odd phase:  Xn ready at its start; reads X for the next even phase behind group 0 and W[g] behind group g.
    // Phase D holds the slice's only barrier: after it every wave has finished READING slice j (its slots are free)
    // and slice j + 1 has landed for everyone (its fragments are read from the next slots during D).
    uint32_t xa[2], wa[2];  // slot-0 LDS addresses of this lane's fragment rows, per k-step
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int coff = ((ks * 4 + fq) ^ (frow & 7)) << 4;
        xa[ks] = lds_addr32(smem) + (wm * (16 * MT) + frow) * RB + coff;
        wa[ks] = lds_addr32(smem) + WBASE + (wn * 64 + frow) * RB + coff;
    }
    u32x4 W[4], X[HB], Xn[HB];
    auto tie = [](u32x4& r) { asm volatile("" : "+v"(r)); };
    // FIRST (k-step 0 of a tile's first slice): the MFMA takes the constant 0 as C, so the accumulators are never
    // cleared -- 128 moves per wave and tile that ran between two tiles with the matrix pipe idle
    auto group = [&](auto gc, auto mhc, u32x4 (&xs_)[HB], auto first_c) {
        constexpr int g = decltype(gc)::value, mh = decltype(mhc)::value;
        constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
        for (int t = 0; t < HB; ++t)
            acc[g][mh * HB + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, W[g]), __builtin_bit_cast(bf16x8, xs_[t]),
                FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[g][mh * HB + t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_x = [&](u32x4 (&dst)[HB], uint32_t base, auto mhc) {
        constexpr int mh = decltype(mhc)::value;
        if (ASRK_ABLATED(2)) return;
        static_for<HB>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            dst[t] = lds_read16<(mh * HB + t) * 16 * RB>(base);
        });
    };
    auto read_w = [&](auto gc, uint32_t base) {
        constexpr int g = decltype(gc)::value;
        if (ASRK_ABLATED(2)) return;
        W[g] = lds_read16<g * 16 * RB>(base);
    };
    auto even_phase = [&](uint32_t xbase, auto first_c) {  // xbase = xa[ks] + X slot offset
        asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
#pragma unroll
        for (int t = 0; t < HB; ++t) tie(X[t]);
        tie(W[0]);
        group(I0{}, I0{}, X, first_c);
        read_x(Xn, xbase, I1{});
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(HB + 2) : "memory");
        tie(W[1]);
        group(I1{}, I0{}, X, first_c);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(HB + 1) : "memory");
        tie(W[2]);
        group(I2{}, I0{}, X, first_c);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(HB) : "memory");
        tie(W[3]);
        group(I3{}, I0{}, X, first_c);
    };
    // `pre` runs between the wait for Xn and the MFMAs (phase D: the barrier); `mid(g)` behind MFMA group g (phase D:
    // part of the next DMA batch each, so that their issue -- 40-75 cycles apiece -- falls under MFMAs in flight
    // instead of all of them following the barrier back to back on both waves of the SIMD at once)
    auto odd_phase = [&](uint32_t xnext, uint32_t wnext, bool reads, auto first_c, auto&& pre, auto&& mid) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < HB; ++t) tie(Xn[t]);
        pre();
        group(I0{}, I1{}, Xn, first_c);
        mid(I0{});
        if (reads) { read_x(X, xnext, I0{}); read_w(I0{}, wnext); }
        group(I1{}, I1{}, Xn, first_c);
        mid(I1{});
        if (reads) read_w(I1{}, wnext);
        group(I2{}, I1{}, Xn, first_c);
        mid(I2{});
        if (reads) read_w(I2{}, wnext);
        group(I3{}, I1{}, Xn, first_c);
        mid(I3{});
        if (reads) read_w(I3{}, wnext);
    };
    auto first_reads = [&](int xs, int ws) {  // fragments of phase A of the slice in X slot xs / W slot ws
        read_x(X, xa[0] + xs * SLOT, I0{});
        read_w(I0{}, wa[0] + ws * SLOT);
        read_w(I1{}, wa[0] + ws * SLOT);
        read_w(I2{}, wa[0] + ws * SLOT);
        read_w(I3{}, wa[0] + ws * SLOT);
    };

    const int nkt = a.K / BK;  // >= 3 (launcher)
    int t = lidx;
    int m0 = (tm_lo + t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
    set_x(m0);
    set_w(n0);
    // A tile's bias, one column per lane: loaded by asm at the start of the tile into a register the compiler knows
    // nothing about (a compiler-visible load inside the K loop became a select + s_waitcnt vmcnt(0) at the top of EVERY
    // slice; hoisted out of it, a vmcnt(0) in the loop preheader behind the previous epilogue's stores).  It is older
    // than everything else the tile issues, so every counted wait of the K loop covers it.
    float bias_reg = 0.f;
    auto load_bias = [&](int n_tile) {
        bias_reg = 0.f;
        if (a.bias) {
            int n = n_tile + wn * 64 + lane;
            n = n < a.N ? n : a.N - 1;  // columns past N are never stored
            const float* p = a.bias + n;
            asm volatile("global_load_dword %0, %1, off" : "+v"(bias_reg) : "v"(p) : "memory");
        }
    };
    load_bias(n0);
#ifdef ASRK_DIAG  // experiment: XCD x starts x * skew * ~1k cycles late, so that the eight XCDs' store bursts do not coincide
    if (a.skew > 0)
        for (int i = 0; i < xcd * a.skew; ++i) __builtin_amdgcn_s_sleep(16);
#endif
    PSTAMP(0);
    // prologue: slices 0, 1 and X of slice 2; the order X0 W0 X1 W1 X2 is the order of the stream
    stage_x(0, 0, I0{}); stage_x(0, 0, I1{});
    stage_w(0, 0, I0{}); stage_w(0, 0, I1{});
    stage_x(1, 1, I0{}); stage_x(1, 1, I1{});
    stage_w(1, 1, I0{}); stage_w(1, 1, I1{});
    stage_x(2, 2, I0{}); stage_x(2, 2, I1{});
    wait_older<2 * GX + GW>();               // my part of slice 0 ...
    asm volatile("s_barrier" ::: "memory");  // ... and everyone's
    int xs = 0, ws = 0;                      // slots of the slice being multiplied
    bool x_behind = true;                    // the youngest GX DMAs in flight are an X batch for the slice after next
    first_reads(0, 0);
#ifdef ASRK_ABLATE
    ablate_on = true;
#endif
#ifdef ASRK_STAMPS
    int tile_no = 0;
#endif
    // The K stream runs across tiles.  Phase D of slice j (kt of this tile) stages W of slice j + 2 into the W slot it
    // leaves and X of slice j + 3 into the X slot it leaves: W comes from the XCD's L2 and one slice of lead covers it;
    // X comes from the memory-side cache or from HBM, hence the third X slot.  Near the end of a tile
    // those slices belong to the next tile.  The epilogue patches alias the X slot the tile's last slice leaves; the X
    // batch that would go there is issued one slice later, at the first barrier of the next tile.
    while (true) {
        const int tnext = t + wgs_per_xcd;
        const bool has_next = tnext < count;
        const int m0n = has_next ? (tm_lo + tnext / tiles_n) * BM : 0;
        const int n0n = has_next ? (tnext % tiles_n) * BN : 0;
        const bool late_x = t != lidx;  // not the first tile: X of slice 2 is still to be staged (see above)
        auto slice = [&](int kt, auto first_c) {
#ifdef ASRK_STAMPS
            if (kt == 0) { if (tile_no == 0) PSTAMP(1); else if (tile_no == 1) PSTAMP(5); }
#endif
            const int xs1 = xs == NXS - 1 ? 0 : xs + 1;
            const uint32_t xso = xs * SLOT, wso = ws * SLOT;
            even_phase(xa[0] + xso, first_c);                                           // A
            odd_phase(xa[1] + xso, wa[1] + wso, true, first_c, [] {}, [](auto) {});     // B
            even_phase(xa[1] + xso, std::false_type{});                                 // C
            // D: W of slice j + 2 -> W slot ws; X of slice j + 3 -> X slot xs (not behind a tile's last slice: the
            // epilogue patches go there); behind the first slice of a later tile also the X batch left out then
            int w_kt = -1, x_kt = -1;
            if (kt + 2 < nkt) w_kt = kt + 2; else if (has_next) w_kt = kt + 2 - nkt;
            if (kt + 3 < nkt) x_kt = kt + 3; else if (has_next && kt + 1 < nkt) x_kt = kt + 3 - nkt;
            const bool extra_x = late_x && kt == 0;
            const int xs_prev = xs == 0 ? NXS - 1 : xs - 1;
            const bool xb_now = x_behind;
            odd_phase(xa[0] + xs1 * SLOT, wa[0] + (ws ^ 1) * SLOT, kt + 1 < nkt, std::false_type{},
                      [&] {
                          // slice j + 1 complete: everything but the X batch issued last, if there is one
                          if (xb_now) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(GX) : "memory");
                          else wait_all_barrier();
                      },
                      [&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if (g == 0) {
                    if (extra_x) { stage_x(2, xs_prev, I0{}); stage_x(2, xs_prev, I1{}); }
                    // the sources switch to the next tile where its first slice is staged (at K = 192 the X switch and
                    // the late X batch of this tile fall into the same phase: the late batch goes first)
                    if (has_next && kt + 2 == nkt && !ASRK_ABLATED(8)) set_w(n0n);
                    if (w_kt >= 0) stage_w(w_kt, ws, I0{});
                } else if (g == 1) {
                    if (w_kt >= 0) stage_w(w_kt, ws, I1{});
                } else if (g == 2) {
                    if (has_next && kt + 3 == nkt && !ASRK_ABLATED(4)) set_x(m0n);
                    if (x_kt >= 0) stage_x(x_kt, xs, I0{});
                } else {
                    if (x_kt >= 0) stage_x(x_kt, xs, I1{});
                }
            });
            x_behind = x_kt >= 0;
            if (kt + 1 < nkt) { xs = xs1; ws ^= 1; }
        };
        slice(0, std::true_type{});
#pragma nounroll
        for (int kt = 1; kt < nkt; ++kt) slice(kt, std::false_type{});
        // xs / ws are still the slots of the last slice: X slot xs holds the patches now
        char* const patch = smem + xs * SLOT + wave * PATCH;
        {
            const uint32_t pa = lds_addr32(patch) + lane * 4;
            // The wait that covers the bias load, tied to its register so the compiler cannot consume the value above
            // it: at most the GX + GW DMAs of the tile's last two phases are younger than the load and still in flight
            // (vector-memory operations complete in issue order), so this never actually stalls.
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(bias_reg) : "n"(GX + GW) : "memory");
            asm volatile("ds_write_b32 %0, %1" ::"v"(pa), "v"(bias_reg) : "memory");
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
        t = tnext;
        m0 = m0n;
        n0 = n0n;
        xs = xs == NXS - 1 ? 0 : xs + 1;
        ws ^= 1;
        // slice 0 of this tile was complete for every wave at the barrier of the last phase D: no barrier here, a wave
        // that leaves its epilogue early starts multiplying while its SIMD partner is still storing
        first_reads(xs, ws);
        load_bias(n0);
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
int launch(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    a.skew = 0;
#ifdef ASRK_DIAG
    static const int skew = [] { const char* e = getenv("ASRK_GEMM_PSKEW"); return e ? atoi(e) : 0; }();
    a.skew = skew;
#endif
    constexpr int BM = 32 * MT;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int per_xcd = ((tiles_m + 7) / 8) * tiles_n;  // tiles of the fullest XCD
    int wgs_per_xcd = device_cus() / 8;
    if (per_xcd < wgs_per_xcd) wgs_per_xcd = per_xcd;
    auto kern = gemm256p_kernel<MT, EPI, OUT_BF16>;
    static std::atomic<uint64_t> lds_set{0};
    if (asrk_max_lds_once((const void*)kern, LDS_BYTES, lds_set) != ASRK_OK) return ASRK_ERR_HIP;
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

// The kernel addresses X and W by 32-bit byte offsets: both must span less than 4 GiB.
bool asrk_gemm256p_fits(const GemmArgs& a) {
    if (a.x_bs < 0 || a.x_rs < 0) return false;
    const int64_t blocks = (a.M - 1) / a.x_rpb;                               // index of the last row block
    const int64_t rows = a.x_rpb - 1 < a.M - 1 ? a.x_rpb - 1 : a.M - 1;        // largest row index inside a block
    const int64_t lim = (int64_t)1 << 31;                                     // in 2-byte elements
    return blocks * a.x_bs + rows * a.x_rs + a.K < lim && (int64_t)a.N * a.K < lim;
}

// Arguments are validated by asrk_gemm_launch (asrk_gemm.hip); needs splits == 1, N % 8 == 0, K % 64 == 0.
int asrk_gemm256p_launch(const GemmArgs& a, bool gelu, bool out_bf16, hipStream_t st) {
    if (a.splits != 1 || (a.N % 8) != 0 || (a.K % BK) != 0 || a.K < 3 * BK) return ASRK_ERR_SHAPE;
    if (!asrk_gemm256p_fits(a)) return ASRK_ERR_SHAPE;
    // Tile height by the schedule: the 192-row tile (MT = 6) pays when the last round of 256-row tiles would be nearly
    // empty (e.g. whisper-small at 16 clips: N = 768, M = 24000), never at the benchmark's shapes.
#ifdef ASRK_DIAG  // diagnostic build only: ASRK_GEMM_MT = 8 or 6 forces the tile height
    static const int forced = [] { const char* e = getenv("ASRK_GEMM_MT"); return e ? atoi(e) : 0; }();
#else
    constexpr int forced = 0;
#endif
    if (forced == 6 || (forced != 8 && schedule_cost(a, 6) < schedule_cost(a, 8))) return dispatch<6>(a, gelu, out_bf16, st);
    return dispatch<8>(a, gelu, out_bf16, st);
}
