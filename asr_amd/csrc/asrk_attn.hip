// Bidirectional multi-head attention forward, head_dim 64, flash-style (scores never leave the CU).
//
// Replaces HF WhisperAttention's attention_interface call for the encoder
// (transformers/models/whisper/modeling_whisper.py:341-351 -> sdpa / :215-238 eager):
// softmax(Q K^T) V with NO mask (the encoder passes None, :636-640) and scaling = 1.0 because q was
// already multiplied by head_dim^-0.5 (:309; folded into the packed q weights).
//
// One workgroup = 4 waves = 128 query rows of one (clip, head); each wave owns 32 query rows.
// K/V tiles of 64 keys stream through a double-buffered LDS ring with global_load_lds.
// S^T = K Q^T is computed "swapped" (keys on MFMA rows, queries on the lanes), so a lane holds
// the scores of ONE query: the row max/sum are in-lane reductions plus one exchange between the
// two half-waves, and the exponentiated tile is already the B operand of O^T += V^T P^T
// (accumulator-as-operand, no LDS round trip for P).  V^T fragments come from the row-major V
// tile through ds_read_b64_tr_b16.  Softmax statistics and accumulators are fp32.
#include "asrk_common.h"
#include "asrk_gemm.h"
#include "asrk_prof.h"

#include <type_traits>

// The wave's issue priority is raised while it feeds the matrix pipe: four waves of four workgroups share a SIMD, and
// by age alone the one that is multiplying gets the issue slots the three in their softmax leave over (round 3, A/B twice
// on one box: attention 5.998 -> 5.91 ms per pipeline-small step with priority 1, the same with 2).
#ifndef ASRK_ATTN_PRIO
#define ASRK_ATTN_PRIO 1
#endif
#if ASRK_ATTN_PRIO > 0
#define ATTN_PRIO_MFMA() __builtin_amdgcn_s_setprio(ASRK_ATTN_PRIO)
#define ATTN_PRIO_VALU() __builtin_amdgcn_s_setprio(0)
#else
#define ATTN_PRIO_MFMA() do { } while (0)
#define ATTN_PRIO_VALU() do { } while (0)
#endif

#ifdef ASRK_ATTN_STAMPS  // diagnostic build only (tools/attn_stamps.py): s_memtime sums per phase of the key loop
__device__ unsigned long long g_attn_stamps[1024 * 8];
extern "C" int asrk_debug_read_attn_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#define ASTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_sum[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif

namespace {

constexpr int HD = 64;       // head_dim (Whisper: always 64)
constexpr int QT = 128;      // query rows per workgroup
constexpr int KT = 64;       // keys per tile
constexpr int TILE_B = KT * HD * 2;  // 8 KiB

__device__ __forceinline__ bf16x4 lds_read_tr16(const char* p) {
    const short4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
    return __builtin_bit_cast(bf16x4, r);
}

// LOG2Q: q arrives multiplied by log2(e) as well (the encoder folds it into W_q and b_q), so scores are in log2
// units and p = 2^(s - ref).  The reference is then subtracted by the MFMA itself: the score accumulators start
// from -ref (a 16-register block that changes only when the reference moves) instead of 0, and the per-score
// multiply-subtract in front of every v_exp_f32 disappears (31 of ~170 vector instructions per key tile).
template <bool LOG2Q>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void attention_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                           int T, int H, int n_pairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K tile | V tile]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5;   // half-wave
    const int l31 = lane & 31;
    // Workgroup -> (clip, head, query block).  Blocks b and b + 8 run on the same XCD; all query blocks of one
    // (clip, head) pair are given to ONE XCD, adjacent in launch order, so they walk the pair's K/V tiles together
    // and every tile after the first reader's comes out of that XCD's L2.  (With query blocks dealt round-robin
    // over the XCDs each XCD fetched its own copy: 2.5 GB of fabric reads per launch against 0.44 GB of q/k/v.)
    const int nqb = (T + QT - 1) / QT;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int pair = (j / nqb) * 8 + xcd;
    if (pair >= n_pairs) return;
    const int qb = j % nqb;
    const int head = pair % H, clip = pair / H;
    const int d = H * HD;
    const int64_t ld = 3 * (int64_t)d;
    const bf16_t* base = qkv + (int64_t)clip * T * ld + head * HD;
    const int q0 = qb * QT + wave * 32;

    // ---- Q^T fragments (MFMA B operand): lane holds Q[q0 + l31][16 s + 8 hh + 0..7]
    bf16x8 qf[4];
    {
        int qr = q0 + l31;
        qr = qr < T ? qr : T - 1;
        const bf16_t* qp = base + (int64_t)qr * ld + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }

    // ---- staging: each thread moves 2 K chunks and 2 V chunks (16 B) per tile
    const int srow = tid >> 3;  // + 32 * i
    const int slot = tid & 7;
    const int wave_lds = wave * 1024;
    // Sources of this thread's chunks: 32-bit byte offsets from the (workgroup-uniform) start of key tile kt, so that
    // the DMA takes a scalar base + one offset register per chunk and no 64-bit lane arithmetic runs per tile.  The
    // ragged last tile clamps its rows (to the clip's last key) and recomputes its offsets, once per workgroup.
    const int ldb = 3 * d * 2;  // bytes per row of qkv
    uint32_t ko[2], vo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = srow + 32 * i;
        ko[i] = (uint32_t)(row * ldb + (d + (slot ^ ((row >> 1) & 7)) * 8) * 2);            // K image: chunk ^ ((row >> 1) & 7)
        vo[i] = (uint32_t)(row * ldb + (2 * d + (slot ^ (((row >> 1) & 1) << 2)) * 8) * 2);  // V image: chunk ^ (((row >> 1) & 1) << 2)
    }
    // The DMA reads through a buffer descriptor of this (clip, head)'s rows: the tile's start is the instruction's scalar
    // offset and the lane's chunk its 32-bit vector offset, so no vector address arithmetic runs per tile.
    const int tile_step = KT * ldb;  // bytes; (T / KT + 1) * tile_step < 2^31 is checked by the launcher
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(base), 0, T * ldb - head * HD * 2, 0x00020000);
    auto dma16 = [&](uint32_t voff, int soff, char* lds) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds), 16, voff, soff, 0, 0);
    };
    auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
        char* kb = smem + buf * (2 * TILE_B);
        char* vb = kb + TILE_B;
        const int soff = kt * tile_step;
        if ((kt + 1) * KT <= T) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                dma16(ko[i], soff, kb + i * 4096 + wave_lds);
                dma16(vo[i], soff, vb + i * 4096 + wave_lds);
            }
        } else {
            const int last = T - 1 - kt * KT;  // last valid row of this tile
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = srow + 32 * i;
                const uint32_t back = (uint32_t)((row > last ? row - last : 0) * ldb);  // rows past the end read the last key
                dma16(ko[i] - back, soff, kb + i * 4096 + wave_lds);
                dma16(vo[i] - back, soff, vb + i * 4096 + wave_lds);
            }
        }
    };

    f32x16 o[2];
    o[0] = f32x16{0.f};
    o[1] = f32x16{0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;
    float nref = 0.f;        // LOG2Q: minus the softmax reference of this lane's query ...
    f32x16 negm;             // ... and the same value in the 16 registers the score MFMAs start from
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = 0.f;
    constexpr float LOG2E = 1.4426950408889634f;

    // fragment read offsets
    // K (A operand of S^T): row = 32 mt + l31, chunk = 2 s + hh, image swizzle (row >> 1) & 7
    const int k_row_off = l31 * 128;
    const int k_sw = (l31 >> 1) & 7;
    // V^T (A operand of PV) through the transposing read: group g = (lane >> 4) & 1, i = lane & 15
    const int vi = lane & 15, vg = (lane >> 4) & 1;

#ifdef ASRK_ATTN_STAMPS
    unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    const int nkt = (T + KT - 1) / KT;
    stage(0, 0);
    const bool wave_active = q0 < T;  // the last query block of a clip may have waves without a single query (T = 1500: one of 48)

    // ---- S^T[mt] = K[32 mt .. +31][:] . Q^T   (keys on rows/registers, queries on lanes).  LOG2Q: the chains start
    // from -ref (negm is the C operand of each chain's first MFMA, never copied).  MASKED: keys past the clip's end -> -inf
    auto scores = [&](const char* kb, int kt, f32x16 (&sacc)[2], auto nmt_c, auto masked_c, auto ref_c) __attribute__((always_inline)) {
        constexpr int NMT = decltype(nmt_c)::value;
        constexpr bool MASKED = decltype(masked_c)::value;
        constexpr bool FROM_REF = decltype(ref_c)::value;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + mt * 4096 + k_row_off + (((2 * s + hh) ^ k_sw) << 4));
                if (s == 0) {
                    f32x16 z;
#pragma unroll
                    for (int i = 0; i < 16; ++i) z[i] = 0.f;
                    sacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0], FROM_REF ? negm : z, 0, 0, 0);
                } else {
                    sacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[mt], 0, 0, 0);
                }
            }
        }
        if (MASKED) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kt * KT + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (kv >= T) sacc[mt][r] = -INFINITY;
                }
        }
    };
    // ---- the exponentials of a tile against the current reference: p -> bf16 B operands of the PV products, row sum
    auto exps = [&](f32x16 (&sacc)[2], bf16x8 (&pf)[2][2], float& psum, auto nmt_c) __attribute__((always_inline)) {
        constexpr int NMT = decltype(nmt_c)::value;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sacc[mt][r]);  // bare v_exp_f32: exp2f() adds a denormal rescale (5 VALU)
                psum += p;
                pf[mt][r >> 3][r & 7] = (bf16_t)p;
            }
    };
    // ---- LOG2Q, the eager form: tile maximum, reference move when it exceeds the reference by more than THR2 (always
    // on the first tile), exponentials.  Lazy rescale: until then p may reach 2^THR2 (harmless: bf16 has fp32's exponent
    // range and a relative rounding error that does not depend on the magnitude) and the 32 accumulator multiplies are
    // skipped for the whole wave.  softmax is invariant to the reference.
    auto softmax_eager_log2 = [&](f32x16 (&sacc)[2], bf16x8 (&pf)[2][2], float& psum, bool first, auto nmt_c) __attribute__((always_inline)) {
        constexpr int NMT = decltype(nmt_c)::value;
        float mx = sacc[0][0];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[mt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        constexpr float THR2 = 8.0f;
        const bool moved = first || mx > THR2;
        if (__builtin_amdgcn_ballot_w64(moved) != 0) {  // wave-uniform
            const float dlt = moved ? mx : 0.f;
            const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-dlt);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[mt][i] -= dlt;
            nref -= dlt;
#pragma unroll
            for (int i = 0; i < 16; ++i) negm[i] = nref;
        }
        exps(sacc, pf, psum, nmt_c);
    };

    // MASKED (compile time) marks the ragged last tile: a run-time test inside one loop body is turned into 31 v_cndmask
    // per tile by if-conversion.  NMT = 1: a last tile with at most 32 keys multiplies only its first half.
    // FAST (LOG2Q only): no tile maximum and no reference at all -- p = 2^s as the scores stand (log2 units).  softmax is
    // invariant to the reference; all the reference has to do is keep 2^(s - ref) inside fp32's exponent range, and
    // bf16 operands / fp32 accumulators hold p = 2^40 with the same RELATIVE accuracy as p <= 1.  So the loop runs
    // without the 20 max instructions, the half-wave exchange, the accumulator block that held -ref (and the 16 moves
    // that copied it every tile): it is bound by vector-ALU cycles (profiles/r3_attn_base_sq_counters.txt: VALU busy
    // 78 % of a SIMD's time, matrix pipe 54 %).  The guard comes once, after the loop: a query whose row sum is not in
    // [2^-100, 2^100), or whose accumulators are not finite, had scores this shortcut cannot hold (|row maximum| beyond
    // ~100 log2 units: never seen on LayerNorm-ed activations, forced in tests/test_kernels_gpu.py); the workgroup
    // votes and, if any lane failed, runs the whole key loop again in the eager form, whose reference follows the
    // running maximum.  Contributions below 2^-126 are flushed, i.e. keys more than 2^-26 below the row's sum.
    auto tile = [&](int kt, auto masked_c, auto nmt_c, auto fast_c, auto buf_c) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value;  // LDS buffer of tile kt = kt & 1: a constant, so that fragment addresses are immediates
        constexpr bool MASKED = decltype(masked_c)::value;
        constexpr int NMT = decltype(nmt_c)::value;
        constexpr bool FAST = decltype(fast_c)::value && LOG2Q;
        ASTAMP(5);  // loop overhead / whatever lies between two tiles
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ASTAMP(0);  // wait for the tile's DMA + barrier
        if (kt + 1 < nkt) stage(kt + 1, BUF ^ 1);
        ASTAMP(1);  // DMA issue
        if (!wave_active) return;  // (wave-uniform) staging and barriers only
        const char* kb = smem + BUF * (2 * TILE_B);
        const char* vb = kb + TILE_B;

        ATTN_PRIO_MFMA();
        f32x16 sacc[2];
        scores(kb, kt, sacc, nmt_c, masked_c, std::integral_constant<bool, LOG2Q && !FAST>{});
        ATTN_PRIO_VALU();
        ASTAMP(2);  // QK^T: fragment reads + 8 MFMAs issued
        float psum = 0.f;
        bf16x8 pf[2][2];  // [mt][k-step s]: registers 8 s .. 8 s + 7
        if (LOG2Q) {
            if (FAST) {
                exps(sacc, pf, psum, nmt_c);  // p = 2^s, reference 0: see below
            } else {
                softmax_eager_log2(sacc, pf, psum, kt == 0, nmt_c);
            }
        } else {
            float mx = sacc[0][0];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[mt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            constexpr float RESCALE_THR = 5.5f;
            const bool moved = mx > m_run + RESCALE_THR;
            const float m_new = moved ? mx : m_run;
            const float mb = m_new * LOG2E;
            if (__builtin_amdgcn_ballot_w64(moved) != 0) {  // wave-uniform
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
            }
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(sacc[mt][r], LOG2E, -mb));
                    psum += p;
                    pf[mt][r >> 3][r & 7] = (bf16_t)p;
                }
            m_run = m_new;
        }
        l_run += psum;
        ASTAMP(3);  // softmax (includes waiting for the score MFMAs)

        // ---- O^T[ht] += V^T[32 ht ..][keys] . P^T   (k order inside a step: 8 (j >> 2) + 4 hh + (j & 3))
        ATTN_PRIO_MFMA();
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int ht = 0; ht < 2; ++ht) {
                    bf16x8 vf;
#pragma unroll
                    for (int rd = 0; rd < 2; ++rd) {
                        const int row = 32 * mt + 16 * s + 4 * hh + 8 * rd + (vi >> 2);
                        const int chunk = 4 * ht + 2 * vg + ((vi & 3) >> 1);
                        const int off = row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4) + (vi & 1) * 8;
                        const bf16x4 t = lds_read_tr16(vb + off);
                        vf[4 * rd + 0] = t[0]; vf[4 * rd + 1] = t[1]; vf[4 * rd + 2] = t[2]; vf[4 * rd + 3] = t[3];
                    }
                    o[ht] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[mt][s], o[ht], 0, 0, 0);
                }
        ATTN_PRIO_VALU();
        ASTAMP(4);  // PV: transposing reads + 8 MFMAs issued
    };
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    const int nfull = (T % KT) != 0 ? nkt - 1 : nkt;
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    auto run = [&](auto fast_c) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_c)::value && LOG2Q;
        int kt = 0;
        if (nfull > 0) {
            if (FAST) tile(0, std::false_type{}, C2{}, std::true_type{}, B0{});
            else tile(0, std::false_type{}, C2{}, std::false_type{}, B0{});  // eager: the first tile sets the reference
            kt = 1;
        }
        for (; kt + 1 < nfull; kt += 2) {  // kt is odd here
            tile(kt, std::false_type{}, C2{}, fast_c, B1{});
            tile(kt + 1, std::false_type{}, C2{}, fast_c, B0{});
        }
        if (kt < nfull) { tile(kt, std::false_type{}, C2{}, fast_c, B1{}); ++kt; }
        if (nfull < nkt) {  // kt == nfull == nkt - 1
            const bool half = T - nfull * KT <= 32;
            if (kt & 1) {
                if (half) tile(kt, std::true_type{}, C1{}, fast_c, B1{});
                else tile(kt, std::true_type{}, C2{}, fast_c, B1{});
            } else {
                if (half) tile(kt, std::true_type{}, C1{}, fast_c, B0{});
                else tile(kt, std::true_type{}, C2{}, fast_c, B0{});
            }
        }
    };
    run(std::true_type{});
    if (LOG2Q) {
        const float l_chk = l_run + __shfl_xor(l_run, 32, 64);
        float fin = 0.f;  // 0 * x is NaN for x = inf / NaN
#pragma unroll
        for (int i = 0; i < 16; ++i) fin = fmaf(o[0][i], 0.f, fmaf(o[1][i], 0.f, fin));
        const bool bad = wave_active && !(l_chk < 1.2676506e30f && l_chk >= 7.8886091e-31f && fin == 0.f);  // 2^100, 2^-100
        if (__builtin_expect(__syncthreads_or(bad ? 1 : 0), 0)) {  // rare: once more, with the reference following the maximum
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; negm[i] = 0.f; }
            l_run = 0.f;
            nref = 0.f;
            stage(0, 0);
            run(std::false_type{});
        }
    }

#ifdef ASRK_ATTN_STAMPS
    if (lane == 0 && blockIdx.x < 1024 && wave == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) g_attn_stamps[blockIdx.x * 8 + i] = st_sum[i];
        g_attn_stamps[blockIdx.x * 8 + 6] = nkt;
    }
#endif
    // ---- normalise and store: lane holds O[q][32 ht + (r & 3) + 8 (r >> 2) + 4 hh]
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + l31;
    if (q < T) {
        bf16_t* op = out + ((int64_t)clip * T + q) * d + head * HD + 4 * hh;
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (bf16_t)(o[ht][4 * g + j] * inv);
                *reinterpret_cast<bf16x4*>(op + 32 * ht + 8 * g) = v;
            }
    }
}

}  // namespace

int asrk_launch_attention(const bf16_t* qkv, bf16_t* out, int B, int T, int H, bool q_log2e, hipStream_t st) {
    if (!qkv || !out) return ASRK_ERR_NULL;
    if (B <= 0 || T <= 0 || H <= 0 || H > 65535 || B > 65535) return ASRK_ERR_SHAPE;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15)) return ASRK_ERR_ALIGN;
    const size_t lds = 4 * TILE_B;
    AsrkProfScope prof(ASRK_PROF_ATTENTION, 4.0 * B * H * (double)T * T * HD, st);
    const long n_pairs = (long)B * H;
    const long blocks = 8 * ((n_pairs + 7) / 8) * ((T + QT - 1) / QT);
    if (blocks > 0x7fffffffL) return ASRK_ERR_SHAPE;
    if ((long)(T + KT) * 3 * H * HD * 2 >= (1L << 31)) return ASRK_ERR_SHAPE;  // 32-bit byte offsets inside one clip's q | k | v rows
    if (q_log2e)
        hipLaunchKernelGGL(attention_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, st, qkv, out, T, H, (int)n_pairs);
    else
        hipLaunchKernelGGL(attention_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, st, qkv, out, T, H, (int)n_pairs);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}

extern "C" int asrk_attention_bf16(const void* qkv, void* out, int B, int T, int H, void* stream) {
    return asrk_launch_attention((const bf16_t*)qkv, (bf16_t*)out, B, T, H, false, (hipStream_t)stream);
}

// Same, for q that already carries the factor log2(e) on top of head_dim^-0.5 (what the encoder feeds: it folds both
// into W_q / b_q); exported so that the log2-domain path, including its reference handling, has a parity test.
extern "C" int asrk_attention_log2_bf16(const void* qkv, void* out, int B, int T, int H, void* stream) {
    return asrk_launch_attention((const bf16_t*)qkv, (bf16_t*)out, B, T, H, true, (hipStream_t)stream);
}
