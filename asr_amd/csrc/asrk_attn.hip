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
    const int64_t tile_step = (int64_t)KT * ldb;  // bytes
    auto stage = [&](int kt, int buf) {
        char* kb = smem + buf * (2 * TILE_B);
        char* vb = kb + TILE_B;
        const char* tb = uniform_ptr(reinterpret_cast<const char*>(base) + kt * tile_step);
        if ((kt + 1) * KT <= T) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                glds16(tb + ko[i], kb + i * 4096 + wave_lds);
                glds16(tb + vo[i], vb + i * 4096 + wave_lds);
            }
        } else {
            const int last = T - 1 - kt * KT;  // last valid row of this tile
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = srow + 32 * i;
                const uint32_t back = (uint32_t)((row > last ? row - last : 0) * ldb);  // rows past the end read the last key
                glds16(tb + (ko[i] - back), kb + i * 4096 + wave_lds);
                glds16(tb + (vo[i] - back), vb + i * 4096 + wave_lds);
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

    const int nkt = (T + KT - 1) / KT;
    stage(0, 0);
    // MASKED (compile time) marks the ragged last tile: a run-time test inside one loop body is turned
    // into 31 v_cndmask per tile by if-conversion
    auto tile = [&](int kt, auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) stage(kt + 1, (kt + 1) & 1);
        const char* kb = smem + (kt & 1) * (2 * TILE_B);
        const char* vb = kb + TILE_B;

        // ---- S^T[mt] = K[32 mt .. +31][:] . Q^T   (keys on rows/registers, queries on lanes)
        f32x16 sacc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (LOG2Q) {
                sacc[mt] = negm;
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[mt][i] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + mt * 4096 + k_row_off + (((2 * s + hh) ^ k_sw) << 4));
                sacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[mt], 0, 0, 0);
            }
        }
        if (MASKED) {  // mask keys past the end of the clip
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kt * KT + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (kv >= T) sacc[mt][r] = -INFINITY;
                }
        }
        // ---- online softmax for this lane's query
        float mx = sacc[0][0];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[mt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // Lazy rescale: the running reference only moves when the tile's maximum exceeds it by more than
        // RESCALE_THR; until then p may reach e^THR (harmless in fp32 / bf16, whose exponent range is fp32's)
        // and the 32 accumulator multiplies are skipped for the whole wave.  softmax is invariant to the
        // reference, so the result differs from eager rescaling only in rounding.
        float psum = 0.f;
        bf16x8 pf[2][2];  // [mt][k-step s]: registers 8 s .. 8 s + 7
        if (LOG2Q) {
            // sacc already holds s - ref (log2 units); the first tile always sets the reference to its maximum
            constexpr float THR2 = 8.0f;
            const bool moved = kt == 0 || mx > THR2;
            if (__builtin_amdgcn_ballot_w64(moved) != 0) {  // wave-uniform
                const float dlt = moved ? mx : 0.f;
                const float alpha = kt == 0 ? 1.0f : __builtin_amdgcn_exp2f(-dlt);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
#pragma unroll
                for (int i = 0; i < 16; ++i) { sacc[0][i] -= dlt; sacc[1][i] -= dlt; }
                nref -= dlt;
#pragma unroll
                for (int i = 0; i < 16; ++i) negm[i] = nref;
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sacc[mt][r]);
                    psum += p;
                    pf[mt][r >> 3][r & 7] = (bf16_t)p;
                }
        } else {
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
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // bare v_exp_f32: exp2f() adds a denormal-range rescale (5 VALU) per call
                    const float p = __builtin_amdgcn_exp2f(fmaf(sacc[mt][r], LOG2E, -mb));
                    psum += p;
                    pf[mt][r >> 3][r & 7] = (bf16_t)p;
                }
            m_run = m_new;
        }
        l_run += psum;

        // ---- O^T[ht] += V^T[32 ht ..][keys] . P^T   (k order inside a step: 8 (j >> 2) + 4 hh + (j & 3))
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
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
    };
    const int nfull = (T % KT) != 0 ? nkt - 1 : nkt;
    for (int kt = 0; kt < nfull; ++kt) tile(kt, std::false_type{});
    if (nfull < nkt) tile(nkt - 1, std::true_type{});

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
