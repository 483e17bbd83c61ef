// Fused log-mel front-end for gfx950.
//
// Replaces HF WhisperFeatureExtractor._torch_extract_fbank_features
// (transformers/models/whisper/feature_extraction_whisper.py:135-168), reached by the reference
// at src/train.py:54-59 and src/train.py:172-180:
//   hann(400) * frame -> 400-pt real DFT (hop 160, center + reflect) -> |.|^2 -> drop last frame
//   -> mel (201 x n_mels, slaney) -> clamp 1e-10 -> log10 -> per-clip max-8 floor -> (x+4)/4.
//
// Kernel 1 (logmel_tile_kernel): persistent workgroups of 4 waves (4 per CU) walk the 24-frame tiles of the batch.  A
//   WAVE is the unit of work: it owns 6 consecutive frames = 3 frame pairs, each pair transformed by 20 lanes as one
//   complex 400-point FFT, 400 = 20 x 20 (asrk_fft400.h): a 20-point DFT per lane in registers (4 x 5 prime-factor, no
//   inner twiddles), a 20 x 20 transpose through a wave-private 4.8 KiB LDS image (ds_write_b32 in, ds_read_b128 out,
//   both conflict-free; a wave's LDS operations execute in order, so there is NO workgroup barrier anywhere in the
//   transform), a second 20-point DFT, and the separation of the two real spectra with the mirror bin fetched by a
//   cross-lane ds_bpermute.  Samples come straight from global memory (each 20-lane group reads 80 contiguous bytes per
//   instruction; the 60 % frame overlap is served by the vector cache), the window from registers (the workgroups are
//   persistent), the twiddle / mel tables from LDS, loaded once per workgroup.  The power spectra go to the same private
//   region, the two frames of a pair side by side, [pair][bin][2]; the mel projection reads them back by chunks of 4
//   bins: lane = (frame pair, one of 20 equally long lists of filters) walks a flat, wave-uniform list of chunks
//   (descriptor, 4 weights, 4 bins x 2 frames: 4 packed FMAs), takes log10 at the end of each filter and puts the two
//   values into the tile's [n_mels][24] staging image.  The four waves meet only around that image (two barriers per
//   tile).  The store pass (whole 96-byte row segments, (x + 4) / 4 applied, per-wave minimum for kernel 2, per-clip
//   maximum in a register) runs one tile late, behind the next tile's sample loads.
//   36-41 KiB of LDS per workgroup and ~100 VGPRs: 16 independent waves per CU, against 10 waves behind 5 barriers per
//   tile in round 1's kernel (counter evidence: profiles/r2_logmel_*, history in DESIGN.md 4.1).
//   With the clips' valid lengths (optional) a tile that lies entirely in zero padding is written as silence without
//   being loaded or transformed.
// Kernel 2 (logmel_finish_kernel): the per-clip floor max(., clip_max - 8), in place, only where a tile needs it (and
//   write-only where the tiles were skipped as padding).
#include "asrk_common.h"
#include "asrk_fft400.h"
#include "asrk_prof.h"

#ifdef ASRK_STAMPS
// Diagnostic build only (tools/build_diag.sh): cycles per phase of the tile kernel, summed over a workgroup's tiles by
// wave 0 (s_memtime with a full wait in front, so a phase is charged with the latencies it exposes).
__device__ unsigned long long g_lm_stamps[1024 * 16];
extern "C" int asrk_debug_read_lmstamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lm_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#define LSTAMP(i)                                                                              \
    do {                                                                                       \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                           \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                    \
        lm_acc[i] += t_now - lm_prev;                                                          \
        lm_prev = t_now;                                                                       \
    } while (0)
#else
#define LSTAMP(i) do { } while (0)
#endif

namespace {

__device__ __forceinline__ unsigned ordered_key(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_value(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// Orders this wave's LDS stores before its following LDS loads of other lanes' data (the hardware runs a wave's LDS
// operations in order; this only stops the compiler from moving them and makes it wait for the stores' data).
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float lane_fetch(int byte_addr, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

// LDS carve-up (bytes): tables | staging [n_mels][24] | 4 wave-private images (the window lives in registers)
constexpr int OFF_TW = 0;                                    // float2[400]
constexpr int OFF_MI = OFF_TW + NFFT * 8;                    // int[MAX_CHUNKS][MEL_LANES]
constexpr int OFF_MW = OFF_MI + MAX_CHUNKS * MEL_LANES * 4;  // float[MAX_NNZ]
constexpr int OFF_STAGE = OFF_MW + MAX_NNZ * 4;              // float[n_mels][TILE_F]
static_assert(OFF_TW % 8 == 0 && OFF_STAGE % 16 == 0, "LDS alignment");
constexpr size_t tile_lds_bytes(int n_mels) {
    return (size_t)OFF_STAGE + (size_t)n_mels * TILE_F * 4 + (size_t)NWAVES * PRIV_FLOATS * 4;
}
static_assert(tile_lds_bytes(MAX_MELS) <= 163840 / 4, "four workgroups per CU");

constexpr int LOGMEL_WPS = 4;  // workgroups per CU (= waves per SIMD)
constexpr int TGROUP = 4;      // neighbouring tiles of a clip dealt to neighbouring workgroups

// Work item -> (clip, tile of the clip).  Row r = item / (4 n_clips) holds tile group r of every clip, four neighbouring
// tiles of a clip side by side (they share their overlapping samples and complete each other's 128-byte output lines in
// L2), the clips rotated by 61 r so that a workgroup meets a different clip in every round.
__device__ __forceinline__ int item_clip_of(int item, int n_clips) {
    const int per_row = TGROUP * n_clips, r = item / per_row;
    return ((item - r * per_row) / TGROUP + 61 * r) % n_clips;
}
__device__ __forceinline__ int item_tile_of(int item, int n_clips) {  // may be >= tiles_per_clip (padding item)
    return (item / (TGROUP * n_clips)) * TGROUP + item % TGROUP;
}
// A tile lies entirely in a clip's zero padding when the smallest sample index any of its frames touches -- reflection
// at either end of the clip included -- is at or after the clip's valid length.
__device__ __forceinline__ bool tile_is_padding(int tile_of_clip, int valid, int n_samples) {
    const int s0 = tile_of_clip * TILE_F * HOP - NFFT / 2, s1 = s0 + (TILE_F - 1) * HOP + NFFT - 1;
    int lo = s0 > 0 ? s0 : 0;
    if (s1 >= n_samples) lo = min(lo, max(2 * (n_samples - 1) - s1, 0));
    return valid <= lo;
}

constexpr int KEY_STRIDE = 32;  // words between the per-clip maximum keys: one 128-byte line each (see store_tile)
constexpr int FIN_TILES = 4;    // tiles per workgroup of the finish kernel

// ROW_ITERS = ceil(n_mels / 42): the store pass is unrolled by it, so that the number of global stores per tile is known
// at compile time.  A wave's vector-memory operations complete in issue order and the next tile's sample loads are
// issued BEFORE this tile's stores; with a countable number of stores the compiler waits for those loads with
// vmcnt(stores + 1) instead of vmcnt(0) -- which made every wave sit out its own stores' write latency once per tile
// (0.14 of 0.49 ms per 256 clips).
template <int ROW_ITERS, bool VEC>
__global__ __launch_bounds__(NTHREADS, LOGMEL_WPS) void logmel_tile_kernel(const float* __restrict__ wave, int n_samples, int n_frames,
                                                                   int n_mels, int tiles_per_clip, int n_tiles,
                                                                   const MelTables* __restrict__ tab, float* __restrict__ out,
                                                                   unsigned* __restrict__ clip_max_key, float* __restrict__ tile_min,
                                                                   int n_chunks, const int* __restrict__ valid_len) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* Ltw = reinterpret_cast<float2*>(smem + OFF_TW);
    int* Lmi = reinterpret_cast<int*>(smem + OFF_MI);
    float* Lmw = reinterpret_cast<float*>(smem + OFF_MW);
    float* stage = reinterpret_cast<float*>(smem + OFF_STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* priv = stage + n_mels * TILE_F + wv * PRIV_FLOATS;

    // ---- tables, once per workgroup
    for (int i = tid; i < NFFT; i += NTHREADS) Ltw[i] = tab->twiddle[i];
    for (int i = tid; i < MAX_CHUNKS * MEL_LANES; i += NTHREADS) Lmi[i] = tab->mel_chunk[i];
    for (int i = tid; i < MAX_NNZ; i += NTHREADS) Lmw[i] = tab->mel_w[i];
    __syncthreads();

    const int g = lane_pair(lane), j = lane_j(lane);
    const int mir = mirror_lane(g, j) * 4;  // ds_bpermute byte address of the mirror lane
    const bool mel_lane = lane < PAIRS * MEL_LANES;  // mel projection: lane = (frame pair g, filter list j); 60..63 idle
    float win[20];  // this lane's window values, for the life of the (persistent) workgroup
#pragma unroll
    for (int n1 = 0; n1 < 20; ++n1) win[n1] = tab->window[20 * n1 + j];

#ifdef ASRK_STAMPS
    unsigned long long lm_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long lm_prev = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    // Work items: see item_clip_of.  Dealt so that concurrently running workgroups (consecutive items) cover MANY clips:
    // with a clip's 125 tiles on 125 neighbouring workgroups, the 500 waves of a clip all merged their maximum into ONE
    // word at the same time (0.14 of 0.49 ms, see store_tile).
    const int n_clips = n_tiles / (TGROUP * ((tiles_per_clip + TGROUP - 1) / TGROUP));
    auto item_clip = [&](int item) { return item_clip_of(item, n_clips); };
    auto item_tile = [&](int item) { return item_tile_of(item, n_clips); };

    // samples of the lane's frame pair of one tile (center=True, reflect padding; torch.stft semantics): n = 20 n1 + j
    auto load_samples = [&](int tile, float (&re)[20], float (&im)[20]) {
        const int clip = item_clip(tile);
        const int tw0 = item_tile(tile) * TILE_F + FPW * wv;  // first frame of this wave
        const float* x = wave + (size_t)clip * n_samples;
        const int sa = (tw0 + 2 * g) * HOP - NFFT / 2 + j;  // sample of (even frame, n1 = 0)
        const int wlo = tw0 * HOP - NFFT / 2, whi = wlo + WAVE_S - 1;
        if (wlo >= 0 && whi < n_samples) {  // wave-uniform: no reflection anywhere under this wave
            const float* pa = x + sa;
#pragma unroll
            for (int n1 = 0; n1 < 20; ++n1) {
                re[n1] = pa[20 * n1];
                im[n1] = pa[HOP + 20 * n1];
            }
        } else {
            auto at = [&](int s) {
                if (s < 0) s = -s;
                if (s >= n_samples) s = 2 * (n_samples - 1) - s;
                s = min(max(s, 0), n_samples - 1);  // frames past the clip end are computed but never stored
                return x[s];
            };
#pragma unroll
            for (int n1 = 0; n1 < 20; ++n1) {
                re[n1] = at(sa + 20 * n1);
                im[n1] = at(sa + HOP + 20 * n1);
            }
        }
    };

    // The store pass of a tile: its [n_mels][24] values from the staging image, (x + 4) / 4 applied, as whole 96-byte row
    // segments (a per-element path when rows are not 16-byte aligned), plus the wave's minimum for the finish kernel and
    // its share of the per-clip maximum.  It is branch-free on purpose -- buffer stores with the out-of-range lanes pointed
    // past the end of the buffer, which the hardware drops -- so that every wave issues exactly ROW_ITERS (x 4) + 1 stores
    // per tile and the compiler can COUNT them: the pass runs at the top of the NEXT tile, right behind that tile's sample
    // loads, and the wait for those loads is then vmcnt(stores), not vmcnt(0).  A wave's vector-memory operations
    // complete in issue order; with the store pass at the end of its own tile every wave sat out the write latency of its
    // stores once per tile in front of its next loads (0.14 of 0.49 ms).
    //
    // The per-clip maximum is kept in a register while the wave stays on one clip and merged into the clip's key (an
    // ordered-integer atomicMax) only when the clip changes and at the end: integer atomics at agent scope execute at the
    // memory side, ONE AFTER THE OTHER PER 64-BYTE LINE at ~12 ns each, and the wave's next loads queue behind them.  One
    // atomic per wave and tile into keys packed 16 to a line cost ~0.1 ms per call whatever the batch (tools/
    // logmel_bsweep.sh: 97 us for 8 clips against 15 us for 1); hence also one key per 128-byte line (KEY_STRIDE).
    float run_max = -INFINITY;
    int run_clip = -1;
    auto flush_max = [&]() {
        if (run_clip >= 0 && lane == 0) atomicMax(clip_max_key + (size_t)run_clip * KEY_STRIDE, ordered_key(run_max));
    };
    const __amdgpu_buffer_rsrc_t rs_min = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(tile_min), 0, tile_min ? n_clips * tiles_per_clip * NWAVES * 4 : 0, 0x00020000);  // none: all dropped
    auto store_tile = [&](int tile_or_none, bool silent = false) {
        // tile_or_none < 0 (a workgroup's first pass): the same instructions with every lane out of range, so that the
        // loop body has ONE instruction stream -- a branch around the stores would bring back the uncountable case
        const bool valid = tile_or_none >= 0;
        const int tile = valid ? tile_or_none : 0;
        const int clip = __builtin_amdgcn_readfirstlane(item_clip(tile));
        const int tt = item_tile(tile);
        const int t0 = tt * TILE_F;
        const int per_clip = n_mels * n_frames;  // < 2^29 elements (host check)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(uniform_ptr(reinterpret_cast<const char*>(out + (size_t)clip * per_clip))), 0, per_clip * 4, 0x00020000);
        const int c4 = tid % 6, r0 = tid / 6;
        const int fr = t0 + 4 * c4;
        float vmax = -INFINITY, vmin = INFINITY;
#pragma unroll
        for (int it = 0; it < ROW_ITERS; ++it) {
            const int r = r0 + 42 * it;
            const bool row_ok = valid && tid < 42 * 6 && r < n_mels;
            const float4 v = *reinterpret_cast<const float4*>(stage + (row_ok ? r : 0) * TILE_F + 4 * c4);
            const int off = (r * n_frames + fr) * 4;
            const float e[4] = {v.x, v.y, v.z, v.w};
            float y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = (e[k] + 4.0f) * 0.25f;
            if (VEC) {  // n_frames % 4 == 0: a float4 is valid or invalid as a whole
                const bool ok = row_ok && fr < n_frames;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, float4{y[0], y[1], y[2], y[3]}), rs,
                                                       ok ? off : 0x7fffffff, 0, 0);
                vmax = ok ? fmaxf(vmax, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w))) : vmax;
                vmin = ok ? fminf(vmin, fminf(fminf(v.x, v.y), fminf(v.z, v.w))) : vmin;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool ok = row_ok && fr + k < n_frames;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[k]), rs, ok ? off + 4 * k : 0x7fffffff, 0, 0);
                    vmax = ok ? fmaxf(vmax, e[k]) : vmax;
                    vmin = ok ? fminf(vmin, e[k]) : vmin;
                }
            }
        }
        vmax = wave_max(vmax);
        vmin = -wave_max(-vmin);
        // this wave's minimum (log10 units, +inf when it stored nothing) of the tile: slot [clip][tile][wave]
        // (a tile skipped as padding is marked with -inf: the finish kernel then knows it is constant and need not read it)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(silent ? -INFINITY : vmin), rs_min,
                                              valid && lane == 0 ? ((clip * tiles_per_clip + tt) * NWAVES + wv) * 4 : 0x7fffffff, 0, 0);
        if (valid) {  // wave-uniform
            if (clip != run_clip) {
                flush_max();
                run_clip = clip;
                run_max = vmax;
            } else {
                run_max = fmaxf(run_max, vmax);
            }
        }
    };

    float re[20], im[20];
    int prev_tile = -1;
    bool prev_silent = false;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (item_tile(tile) >= tiles_per_clip) continue;  // padding item of the last group of four (the whole workgroup skips)
        // A tile that lies entirely in a clip's zero padding (the caller passed the clips' valid lengths): every frame is
        // exactly silence, i.e. log10(1e-10) = -10 in every filter -- what the full computation gives bit for bit -- so
        // the samples are neither loaded nor transformed.  (Workgroup-uniform.  The rotation of the clips over the rounds
        // evens the clips' lengths out over the workgroups; an exact balance by a compacted work list was measured: its
        // one-workgroup scan costs 49 us per call, more than the 16 us it takes off the tile kernel at 256 clips.)
        if (valid_len && tile_is_padding(item_tile(tile), valid_len[item_clip(tile)], n_samples)) {
            store_tile(prev_tile, prev_silent);
            prev_tile = tile;
            prev_silent = true;
            __syncthreads();  // staging free
            for (int i = lane; i < n_mels * FPW; i += 64) stage[(i / FPW) * TILE_F + FPW * wv + i % FPW] = -10.0f;
            __syncthreads();  // staging complete
            continue;
        }
        LSTAMP(9);  // loop overhead / previous tile's tail
        load_samples(tile, re, im);
        store_tile(prev_tile, prev_silent);  // behind this tile's loads: see store_tile
        prev_tile = tile;
        prev_silent = false;
        LSTAMP(7);  // store pass (of the previous tile)
        LSTAMP(0);  // sample loads (what is left exposed of them)
        fft400_stage1(re, im, win, Ltw, j);
        LSTAMP(1);  // window, DFT 1, twiddle

        // ---- 20 x 20 transpose inside the pair, real parts then imaginary parts through the same private image
        float zr[20], zi[20];
        float* xw = priv + xch_index(g, 0, j);
        const float4* xr = reinterpret_cast<const float4*>(priv + xch_index(g, j, 0));
#pragma unroll
        for (int k1 = 0; k1 < 20; ++k1) xw[k1 * 20] = re[k1];
        wave_lds_order();
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const float4 v = xr[c];
            zr[4 * c] = v.x; zr[4 * c + 1] = v.y; zr[4 * c + 2] = v.z; zr[4 * c + 3] = v.w;
        }
        wave_lds_order();
#pragma unroll
        for (int k1 = 0; k1 < 20; ++k1) xw[k1 * 20] = im[k1];
        wave_lds_order();
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const float4 v = xr[c];
            zi[4 * c] = v.x; zi[4 * c + 1] = v.y; zi[4 * c + 2] = v.z; zi[4 * c + 3] = v.w;
        }
        LSTAMP(2);  // exchange
        dft20(zr, zi);  // Z[j + 20 k2] in register k2
        LSTAMP(3);  // DFT 2

        // ---- separate the two real spectra; |X|^2 of bins 0..200 into the private image [pair][bin][2 frames]
        {
            float2* pp = reinterpret_cast<float2*>(priv + g * PPAIR) + j;
#pragma unroll
            for (int k2 = 0; k2 < 10; ++k2) {
                float cr = lane_fetch(mir, zr[mirror_reg(k2)]), ci = lane_fetch(mir, zi[mirror_reg(k2)]);
                if (j == 0) { cr = zr[mirror_reg_lane0(k2)]; ci = zi[mirror_reg_lane0(k2)]; }
                float2 pw;
                fft400_power(zr[k2], zi[k2], cr, ci, pw.x, pw.y);
                pp[20 * k2] = pw;
            }
            if (j == 0) {  // bin 200: its mirror is itself
                float2 pw;
                fft400_power(zr[10], zi[10], zr[mirror_reg_lane0(10)], zi[mirror_reg_lane0(10)], pw.x, pw.y);
                pp[200] = pw;
            } else if (j < 4) {  // bins 201..203 of the last chunk: zeros (the exchange image leaves numbers there)
                pp[200] = make_float2(0.f, 0.f);
            }
        }
        LSTAMP(4);  // mirror + power
        __syncthreads();  // every wave has stored the previous tile: the staging image is free (and orders the P image)
        LSTAMP(8);  // barrier: staging free

        // ---- mel projection + log10: lane = (frame pair g, filter list j), one flat list of chunks (4 bins x 4 weights,
        //      both frames of the pair) per lane.  Two chunks at a time, all their loads first, so that one LDS round trip
        //      serves 16 products (per-filter loops ran one dependent round trip per chunk: 17 k cycles per tile).
        if (mel_lane) {
            const char* prow = reinterpret_cast<const char*>(priv + g * PPAIR);
            const char* wbase = reinterpret_cast<const char*>(Lmw);
            char* srow = reinterpret_cast<char*>(stage + FPW * wv + 2 * g);
            const int* dq = Lmi + j;
            f32x2_t acc = f32x2_t{0.f, 0.f};
            // Bank spreading: the weights lie [step][list], so the 20 lists read 320 consecutive bytes; the two 16-byte
            // halves of a power chunk (bins 4c, 4c+1 | 4c+2, 4c+3) are read in opposite order by odd lists -- whose weight
            // chunks are stored with their halves swapped to match -- so that each ds_read_b128 touches both halves of the
            // 32-byte chunk grid and not only every other group of four banks.
            const int hsw = (j & 1) << 4;
            for (int e0 = 0; e0 < n_chunks; e0 += 2) {  // n_chunks: even, the same for every lane
                int d[2];
                float4 w[2], qa[2], qb[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) d[u] = dq[(e0 + u) * MEL_LANES];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const char* pc = prow + ((d[u] & 63) << 5);
                    qa[u] = *reinterpret_cast<const float4*>(pc + hsw);         // even lists: bins 4c, 4c+1 x (even, odd frame)
                    qb[u] = *reinterpret_cast<const float4*>(pc + (hsw ^ 16));  // even lists: bins 4c+2, 4c+3
                    w[u] = *reinterpret_cast<const float4*>(wbase + ((e0 + u) * MEL_LANES + j) * 16);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc += f32x2_t{qa[u].x, qa[u].y} * w[u].x;
                    acc += f32x2_t{qa[u].z, qa[u].w} * w[u].y;
                    acc += f32x2_t{qb[u].x, qb[u].y} * w[u].z;
                    acc += f32x2_t{qb[u].z, qb[u].w} * w[u].w;
                    if (d[u] & CHUNK_LAST) {
                        // clamp(min=1e-10).log10(): silence is exactly -10.  v_log_f32 (log2, 1 ulp) * log10(2):
                        // |err| < 2e-6 here, against ~25 instructions for log10f()
                        float2 o;
                        o.x = acc.x > 1e-10f ? __builtin_amdgcn_logf(acc.x) * 0.30102999566398120f : -10.0f;
                        o.y = acc.y > 1e-10f ? __builtin_amdgcn_logf(acc.y) * 0.30102999566398120f : -10.0f;
                        *reinterpret_cast<float2*>(srow + ((d[u] >> 14) & 127) * (TILE_F * 4)) = o;
                        acc = f32x2_t{0.f, 0.f};
                    }
                }
            }
        }

        LSTAMP(5);  // mel
        __syncthreads();  // the staging image is complete; it is stored at the top of the next tile (store_tile)
        LSTAMP(6);  // barrier: staging complete
    }
    if (prev_tile >= 0) store_tile(prev_tile, prev_silent);
    flush_max();
#ifdef ASRK_STAMPS
    if (tid == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 10; ++i) g_lm_stamps[blockIdx.x * 16 + i] = lm_acc[i];
#endif
}

using tile_kernel_t = void (*)(const float*, int, int, int, int, int, const MelTables*, float*, unsigned*, float*, int, const int*);
// vec: rows of the output are 16-byte aligned (n_frames % 4 == 0): float4 stores
tile_kernel_t tile_kernel_for(int n_mels, bool vec) {
    switch ((n_mels + 41) / 42) {
        case 1: return vec ? logmel_tile_kernel<1, true> : logmel_tile_kernel<1, false>;
        case 2: return vec ? logmel_tile_kernel<2, true> : logmel_tile_kernel<2, false>;  // 80 mel bins
        case 3: return vec ? logmel_tile_kernel<3, true> : logmel_tile_kernel<3, false>;
        case 4: return vec ? logmel_tile_kernel<4, true> : logmel_tile_kernel<4, false>;  // 128 mel bins
        default: return nullptr;
    }
}

// Kernel 2: the per-clip floor max(y, ((clip_max - 8) + 4) / 4) in place -- the same value, bit for bit, as
// (max(x, clip_max - 8) + 4) / 4 (HF :163-165), since x -> (x + 4) / 4 is monotone.  One workgroup per (clip, group of four
// tiles = 96 frames = 384-byte row segments); it leaves at once when none of the 16 wave minima of its tiles is below
// the floor, which is every group of a clip without silence -- then this pass reads 64 bytes per group instead of
// rewriting the features (0.07 of 0.43 ms per 256 clips).  tile_min == nullptr (the no-alloc entry point): no shortcut.
template <bool VEC4>
__global__ __launch_bounds__(256) void logmel_finish_kernel(float* __restrict__ out, const unsigned* __restrict__ key,
                                                             const float* __restrict__ tile_min, int n_mels, int n_frames,
                                                             int tiles_per_clip) {
    const int clip = blockIdx.y, grp = blockIdx.x;
    const float floor_x = ordered_value(key[(size_t)clip * KEY_STRIDE]) - 8.0f;
    bool fill = false;  // every tile of the group was skipped as padding: constant -10, nothing to read
    if (tile_min) {
        const int t_lo = grp * FIN_TILES, t_hi = min(t_lo + FIN_TILES, tiles_per_clip);
        float m = INFINITY, mm = -INFINITY;
        for (int i = t_lo * NWAVES + (int)threadIdx.x; i < t_hi * NWAVES; i += 256) {
            const float v = tile_min[(size_t)clip * tiles_per_clip * NWAVES + i];
            m = fminf(m, v);
            mm = fmaxf(mm, v);
        }
        if (!__syncthreads_or(m < floor_x)) return;
        fill = !__syncthreads_or(mm != -INFINITY);
    }
    const float floor_y = (floor_x + 4.0f) * 0.25f;
    const float fill_y = fmaxf((-10.0f + 4.0f) * 0.25f, floor_y);
    float* o = out + (size_t)clip * n_mels * n_frames;
    const int f_lo = grp * FIN_TILES * TILE_F, f_hi = min(f_lo + FIN_TILES * TILE_F, n_frames);
    if (VEC4) {  // n_frames % 4 == 0: rows and segments are 16-byte aligned
        const int w4 = (f_hi - f_lo) >> 2;  // float4 per row segment (24 for a whole group)
        for (int i = threadIdx.x; i < n_mels * w4; i += 256) {
            const int r = i / w4, c = i - r * w4;
            float4* q = reinterpret_cast<float4*>(o + (size_t)r * n_frames + f_lo) + c;
            if (fill) {
                *q = float4{fill_y, fill_y, fill_y, fill_y};
                continue;
            }
            float4 v = *q;
            v.x = fmaxf(v.x, floor_y); v.y = fmaxf(v.y, floor_y); v.z = fmaxf(v.z, floor_y); v.w = fmaxf(v.w, floor_y);
            *q = v;
        }
    } else {
        const int w = f_hi - f_lo;
        for (int i = threadIdx.x; i < n_mels * w; i += 256) {
            const int r = i / w, c = i - r * w;
            float* q = o + (size_t)r * n_frames + f_lo + c;
            *q = fill ? fill_y : fmaxf(*q, floor_y);
        }
    }
}

}  // namespace

// Scratch of asrk_logmel_f32_noalloc: the per-clip maxima (ordered-integer keys), one 128-byte line per clip (512 KiB).  In host code this
// name is only a stand-in for the symbol, not a device address: every device has its own copy at its own address, which
// asrk_logmel_plan_create looks up once (hipGetSymbolAddress, with the plan's device current) and keeps in the plan;
// kernels and hipMemsetAsync only ever get that pointer.
static __device__ __align__(256) unsigned g_noalloc_clip_max[ASRK_LOGMEL_NOALLOC_MAX_CLIPS * KEY_STRIDE];

struct asrk_logmel_plan {
    int n_mels;
    int device;
    int n_cus;
    int n_chunks;               // chunks per mel lane (even)
    MelTables* tables;          // device
    unsigned* noalloc_scratch;  // this device's address of g_noalloc_clip_max
};

extern "C" int asrk_logmel_plan_create(int n_mels, const float* mel_filters, asrk_logmel_plan** plan) {
    if (!mel_filters || !plan) return ASRK_ERR_NULL;
    if (n_mels < 1 || n_mels > MAX_MELS) return ASRK_ERR_SHAPE;
    MelTables* h = new MelTables();
    const double two_pi = 6.283185307179586476925286766559;
    for (int n = 0; n < NFFT; ++n) h->window[n] = (float)(0.5 * (0.5 - 0.5 * cos(two_pi * n / NFFT)));  // Hann / 2 (asrk_fft400.h)
    for (int k1 = 0; k1 < 20; ++k1)
        for (int n2 = 0; n2 < 20; ++n2) {
            const double a = -two_pi * (double)(n2 * k1) / NFFT;
            h->twiddle[k1 * 20 + n2] = make_float2((float)cos(a), (float)sin(a));
        }
    // sparse rows of the bank cut into 16-byte chunks on the bin grid; the filters dealt to the MEL_LANES lists longest
    // first onto the shortest list (asrk_fft400.h)
    for (int i = 0; i < MAX_NNZ; ++i) h->mel_w[i] = 0.f;  // mel_w[step][list][4]; padding chunks keep zero weights
    for (int i = 0; i < MAX_CHUNKS * MEL_LANES; ++i) h->mel_chunk[i] = 0;  // padding chunk: bins 0..3, no output
    int flo[MAX_MELS], fnch[MAX_MELS], order[MAX_MELS];
    for (int m = 0; m < n_mels; ++m) {
        int lo = NBINS, hi = -1;
        for (int k = 0; k < NBINS; ++k)
            if (mel_filters[(size_t)k * n_mels + m] != 0.0f) { lo = k < lo ? k : lo; hi = k; }
        if (hi < 0) { lo = 0; hi = 0; }  // an empty filter: one chunk of zero weights, so that it still emits log10(1e-10)
        flo[m] = lo / 4;
        fnch[m] = hi / 4 - lo / 4 + 1;
        order[m] = m;
    }
    for (int a = 1; a < n_mels; ++a)  // insertion sort by chunk count, descending, stable
        for (int b = a; b > 0 && fnch[order[b]] > fnch[order[b - 1]]; --b) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
    int per_lane[MEL_LANES] = {0};
    for (int i = 0; i < n_mels; ++i) {
        const int m = order[i], c0 = flo[m], nch = fnch[m];
        int q = 0;
        for (int l = 1; l < MEL_LANES; ++l)
            if (per_lane[l] < per_lane[q]) q = l;
        if (per_lane[q] + nch > MAX_CHUNKS) { delete h; return ASRK_ERR_SHAPE; }
        for (int c = 0; c < nch; ++c) {
            const int e = per_lane[q] + c;
            h->mel_chunk[e * MEL_LANES + q] = (c0 + c) | (m << 14) | (c == nch - 1 ? CHUNK_LAST : 0);
            for (int k = 0; k < 4; ++k) {
                const int bin = 4 * (c0 + c) + k;
                const float wv = bin < NBINS ? mel_filters[(size_t)bin * n_mels + m] : 0.f;
                h->mel_w[(e * MEL_LANES + q) * 4 + ((q & 1) ? (k ^ 2) : k)] = wv;  // odd lists: halves swapped (kernel)
            }
        }
        per_lane[q] += nch;
    }
    int n_chunks = 0;
    for (int q = 0; q < MEL_LANES; ++q) n_chunks = per_lane[q] > n_chunks ? per_lane[q] : n_chunks;
    n_chunks = (n_chunks + 1) / 2 * 2;
    if (n_chunks > MAX_CHUNKS) { delete h; return ASRK_ERR_SHAPE; }
    asrk_logmel_plan* p = new asrk_logmel_plan();
    p->n_mels = n_mels;
    p->n_chunks = n_chunks;
    p->n_cus = 256;
    if (hipGetDevice(&p->device) != hipSuccess ||
        hipDeviceGetAttribute(&p->n_cus, hipDeviceAttributeMultiprocessorCount, p->device) != hipSuccess || p->n_cus < 1 ||
        hipMalloc(&p->tables, sizeof(MelTables)) != hipSuccess ||
        hipMemcpy(p->tables, h, sizeof(MelTables), hipMemcpyHostToDevice) != hipSuccess) {
        delete h;
        delete p;
        return ASRK_ERR_HIP;
    }
    delete h;
    // one-time set-up for the entry point that allocates nothing: this device's address of the static scratch array, and
    // the tile kernel's LDS opt-in (so that the first asrk_logmel_f32_noalloc call changes no function attribute either)
    void* sym = nullptr;
    static std::atomic<uint64_t> lds_set[2][4];
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_noalloc_clip_max)) != hipSuccess || !sym ||
        !tile_kernel_for(n_mels, true) ||
        asrk_max_lds_once((const void*)tile_kernel_for(n_mels, true), (int)tile_lds_bytes(MAX_MELS), lds_set[0][(n_mels + 41) / 42 - 1]) != ASRK_OK ||
        asrk_max_lds_once((const void*)tile_kernel_for(n_mels, false), (int)tile_lds_bytes(MAX_MELS), lds_set[1][(n_mels + 41) / 42 - 1]) != ASRK_OK) {
        (void)hipFree(p->tables);
        delete p;
        return ASRK_ERR_HIP;
    }
    p->noalloc_scratch = static_cast<unsigned*>(sym);
    *plan = p;
    return ASRK_OK;
}

extern "C" void asrk_logmel_plan_destroy(asrk_logmel_plan* plan) {
    if (!plan) return;
    (void)hipFree(plan->tables);
    delete plan;
}

namespace {
// Both entry points: clear the per-clip maxima, tile kernel, finish kernel.  `prof`: bracket with events when the
// measurement hook is on (event creation allocates, so the no-alloc entry point never does).
int logmel_launch(const asrk_logmel_plan* plan, const float* wave, const int* valid_len, int B, int n_samples, float* out,
                  unsigned* clip_max_key, float* tile_min, bool prof_ok, hipStream_t st) {
    const int n_frames = n_samples / HOP;
    const auto kern = tile_kernel_for(plan->n_mels, (n_frames % 4) == 0);  // LDS opt-in: once per kernel, in plan_create
    if (!kern) return ASRK_ERR_SHAPE;
    ASRK_CHECK_HIP(hipMemsetAsync(clip_max_key, 0, (size_t)B * KEY_STRIDE * 4, st));
    // algorithmic bytes: read the waveform once, write the features once
    AsrkProfScope prof(prof_ok ? ASRK_PROF_LOGMEL : -1, (double)B * ((double)n_samples * 4 + (double)plan->n_mels * n_frames * 4), st);
    const int tiles_per_clip = (n_frames + TILE_F - 1) / TILE_F;
    const long n_tiles = (long)((tiles_per_clip + 3) / 4 * 4) * B;  // work items: groups of four tiles (the kernel's TGROUP)
    if (n_tiles > 0x7fffffffL / (NWAVES * 4)) return ASRK_ERR_SHAPE;
    // persistent workgroups: 4 per CU (LDS and registers allow exactly that), each walks tiles b, b + grid, ...
    const long resident = (long)LOGMEL_WPS * plan->n_cus;
    const int grid = (int)(n_tiles < resident ? n_tiles : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), tile_lds_bytes(plan->n_mels), st, wave, n_samples, n_frames,
                       plan->n_mels, tiles_per_clip, (int)n_tiles, plan->tables, out, clip_max_key, tile_min, plan->n_chunks,
                       valid_len);
    ASRK_CHECK_HIP(hipGetLastError());
    const dim3 fgrid((tiles_per_clip + FIN_TILES - 1) / FIN_TILES, B);
    if ((n_frames & 3) == 0)
        hipLaunchKernelGGL(logmel_finish_kernel<true>, fgrid, dim3(256), 0, st, out, clip_max_key, tile_min, plan->n_mels, n_frames,
                           tiles_per_clip);
    else
        hipLaunchKernelGGL(logmel_finish_kernel<false>, fgrid, dim3(256), 0, st, out, clip_max_key, tile_min, plan->n_mels, n_frames,
                           tiles_per_clip);
    ASRK_CHECK_HIP(hipGetLastError());
    return ASRK_OK;
}

// workspace of asrk_logmel_f32: keys [B][KEY_STRIDE] | tile minima [B][tiles_per_clip][NWAVES]
size_t ws_keys_bytes(int B) { return ((size_t)B * KEY_STRIDE * 4 + 255) / 256 * 256; }

int logmel_check(const asrk_logmel_plan* plan, const float* wave, int B, int n_samples, const float* out) {
    if (!plan || !wave || !out) return ASRK_ERR_NULL;
    if (B < 0 || n_samples <= NFFT / 2) return ASRK_ERR_SHAPE;
    if (((uintptr_t)out & 15) || ((uintptr_t)wave & 3)) return ASRK_ERR_ALIGN;
    if (n_samples / HOP < 1 || B > 65535) return ASRK_ERR_SHAPE;
    if ((long)plan->n_mels * (n_samples / HOP) >= (1L << 29)) return ASRK_ERR_SHAPE;  // 32-bit byte offsets inside a clip's features
    // the plan's tables (and the no-alloc scratch) live on the device that was current in asrk_logmel_plan_create: a launch
    // from another device would hand its kernels the wrong GPU's addresses
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return ASRK_ERR_HIP;
    if (dev != plan->device) return ASRK_ERR_UNSUPPORTED;
    return ASRK_OK;
}
}  // namespace

extern "C" size_t asrk_logmel_workspace_bytes(int B, int n_samples) {
    if (B <= 0 || n_samples < HOP) return 0;
    const int tiles_per_clip = (n_samples / HOP + TILE_F - 1) / TILE_F;
    return ws_keys_bytes(B) + (size_t)B * tiles_per_clip * NWAVES * 4;
}

extern "C" int asrk_logmel_f32(const asrk_logmel_plan* plan, const float* wave, const int* valid_samples, int B, int n_samples,
                               float* out, void* ws, size_t ws_bytes, void* stream) {
    const int rc = logmel_check(plan, wave, B, n_samples, out);
    if ((uintptr_t)valid_samples & 3) return ASRK_ERR_ALIGN;
    if (rc != ASRK_OK) return rc;
    if (B == 0) return ASRK_OK;
    if (!ws) return ASRK_ERR_NULL;
    if ((uintptr_t)ws & 255) return ASRK_ERR_ALIGN;
    if (ws_bytes < asrk_logmel_workspace_bytes(B, n_samples)) return ASRK_ERR_WORKSPACE;
    return logmel_launch(plan, wave, valid_samples, B, n_samples, out, static_cast<unsigned*>(ws),
                         reinterpret_cast<float*>(static_cast<char*>(ws) + ws_keys_bytes(B)), true, (hipStream_t)stream);
}

extern "C" int asrk_logmel_f32_noalloc(const asrk_logmel_plan* plan, const float* wave, int B, int n_samples, float* out,
                                       void* stream) {
    const int rc = logmel_check(plan, wave, B, n_samples, out);
    if (rc != ASRK_OK) return rc;
    if (B > ASRK_LOGMEL_NOALLOC_MAX_CLIPS) return ASRK_ERR_WORKSPACE;  // beyond the static scratch: use asrk_logmel_f32
    if (!plan->noalloc_scratch) return ASRK_ERR_UNSUPPORTED;  // (logmel_check has compared the plan's device with the current one)
    if (B == 0) return ASRK_OK;
    // no tile minima (they would need 2 KiB per clip of scratch): the finish kernel then visits every group
    return logmel_launch(plan, wave, nullptr, B, n_samples, out, plan->noalloc_scratch, nullptr, false, (hipStream_t)stream);
}
