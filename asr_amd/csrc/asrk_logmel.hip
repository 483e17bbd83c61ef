// Fused log-mel front-end for gfx950.
//
// Replaces HF WhisperFeatureExtractor._torch_extract_fbank_features
// (transformers/models/whisper/feature_extraction_whisper.py:135-168), reached by the reference
// at src/train.py:54-59 and src/train.py:172-180:
//   hann(400) * frame -> 400-pt real DFT (hop 160, center + reflect) -> |.|^2 -> drop last frame
//   -> mel (201 x n_mels, slaney) -> clamp 1e-10 -> log10 -> per-clip max-8 floor -> (x+4)/4.
//
// Kernel 1 (logmel_tile_kernel): one workgroup = 320 threads = 16 groups of 20 lanes, one tile of
//   32 frames of one clip.  The tile's 5360 samples are read from HBM once (coalesced) into LDS.
//   Each 20-lane group transforms TWO real frames at once as one 400-point complex FFT
//   (z = x_even_frame + i * x_odd_frame), factored 400 = 20 x 20: every lane runs a 20-point DFT
//   entirely in registers (4 x 5 prime-factor, no internal twiddles), the groups exchange through
//   LDS once, and the two spectra are separated with X1[k] = (Z[k] + conj Z[400-k]) / 2,
//   X2[k] = (Z[k] - conj Z[400-k]) / 2i.  Power spectra go to LDS, the mel projection uses the
//   sparse rows of the filter bank (each FFT bin feeds <= 2 filters), and log10 values are written
//   as [clip][mel][frame] with frames on the lanes (coalesced).  The per-clip maximum is reduced
//   per workgroup and merged with one ordered-integer atomicMax.
// Kernel 2 (logmel_finish_kernel): applies max(x, clip_max - 8) and (x + 4) / 4 in place.
#include "asrk_common.h"
#include "asrk_fft400.h"
#include "asrk_prof.h"

namespace {

__device__ __forceinline__ unsigned ordered_key(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_value(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__global__ __launch_bounds__(NTHREADS) void logmel_tile_kernel(const float* __restrict__ wave, int n_samples,
                                                                int n_frames, int n_mels,
                                                                const MelTables* __restrict__ tab,
                                                                float* __restrict__ out,
                                                                unsigned* __restrict__ clip_max_key) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* S = reinterpret_cast<float*>(smem);                                  // [TILE_S]
    float2* Y = reinterpret_cast<float2*>(smem + TILE_S * 4);                   // [NGROUPS][YGROUP]
    float2* E = Y;                                                              // [NGROUPS][20][EROW] (aliases Y)
    float* P = reinterpret_cast<float*>(smem + TILE_S * 4 + NGROUPS * 20 * EROW * 8);  // [TILE_F][PSTRIDE]
    // CSR mel rows after the P image
    char* csr = smem + TILE_S * 4 + NGROUPS * 20 * EROW * 8 + TILE_F * PSTRIDE * 4;
    int* Lstart = reinterpret_cast<int*>(csr);
    int* Lptr = Lstart + MAX_MELS;
    float* Lw = reinterpret_cast<float*>(Lptr + MAX_MELS + 1);
    float* Lred = Lw + MAX_NNZ;  // [8] block reduction

    const int tid = threadIdx.x;
    const int clip = blockIdx.y;
    const int t0 = blockIdx.x * TILE_F;  // first frame of the tile
    const int g = tid / 20, j = tid % 20;
    const float* x = wave + (size_t)clip * n_samples;

    // ---- stage the tile's samples (center=True, reflect padding; torch.stft semantics)
    const int s0 = t0 * HOP - NFFT / 2;
    // interior tiles (no reflection, 16-byte aligned: t0 is a multiple of 32, so s0 is a multiple of 4 samples)
    // copy with 16-byte loads; only a clip's first and last tiles take the indexed path
    const bool interior = s0 >= 0 && s0 + TILE_S <= n_samples && ((n_samples & 3) == 0) && ((reinterpret_cast<uintptr_t>(wave) & 15) == 0);
    if (interior) {
        const float4* src = reinterpret_cast<const float4*>(x + s0);
        for (int i = tid; i < TILE_S / 4; i += NTHREADS) reinterpret_cast<float4*>(S)[i] = src[i];
    } else
    for (int i = tid; i < TILE_S; i += NTHREADS) {
        int s = s0 + i;
        if (s < 0) s = -s;
        if (s >= n_samples) s = 2 * (n_samples - 1) - s;
        s = min(max(s, 0), n_samples - 1);  // frames past the clip end are computed but never stored
        S[i] = x[s];
    }
    for (int i = tid; i < n_mels; i += NTHREADS) { Lstart[i] = tab->mel_start[i]; Lptr[i] = tab->mel_ptr[i]; }
    if (tid == 0) Lptr[n_mels] = tab->mel_ptr[n_mels];
    const int nnz = tab->mel_ptr[n_mels];
    for (int i = tid; i < nnz; i += NTHREADS) Lw[i] = tab->mel_w[i];
    __syncthreads();

    fft400_stage1(S, Y, tab, g, j);
    __syncthreads();
    float zr[20], zi[20];
    fft400_stage2(Y, g, j, zr, zi);
    __syncthreads();  // everyone is done reading Y before E (aliasing Y) is written
    fft400_publish(E, g, j, zr, zi);
    __syncthreads();
    fft400_power(E, P, g, j, zr, zi);
    __syncthreads();

    // ---- mel projection (sparse rows) + log10, frames on the lanes
    const int f = tid & 31;
    const bool valid = (t0 + f) < n_frames;
    float vmax = -INFINITY;
    float* obase = out + (size_t)clip * n_mels * n_frames + t0 + f;
    const float* prow = P + f * PSTRIDE;
    for (int m = tid >> 5; m < n_mels; m += NTHREADS / 32) {
        const int b0 = Lstart[m], p0 = Lptr[m], p1 = Lptr[m + 1];
        float acc = 0.f;
        for (int q = p0; q < p1; ++q) acc = fmaf(Lw[q], prow[b0 + q - p0], acc);
        // clamp(min=1e-10).log10(): silence is exactly -10.  v_log_f32 (log2, 1 ulp) * log10(2): |err| < 2e-6 here,
        // against ~25 instructions for log10f()
        const float v = acc > 1e-10f ? __builtin_amdgcn_logf(acc) * 0.30102999566398120f : -10.0f;
        if (valid) {
            obase[(size_t)m * n_frames] = v;
            vmax = fmaxf(vmax, v);
        }
    }
    vmax = wave_max(vmax);
    if ((tid & 63) == 0) Lred[tid >> 6] = vmax;
    __syncthreads();
    if (tid == 0) {
        float v = Lred[0];
#pragma unroll
        for (int i = 1; i < NTHREADS / 64; ++i) v = fmaxf(v, Lred[i]);
        if (v > -INFINITY) atomicMax(clip_max_key + clip, ordered_key(v));
    }
}

template <bool VEC4>
__global__ __launch_bounds__(256) void logmel_finish_kernel(float* __restrict__ out, const unsigned* __restrict__ key,
                                                             int per_clip) {
    const int clip = blockIdx.y;
    const float floor_v = ordered_value(key[clip]) - 8.0f;
    float* o = out + (size_t)clip * per_clip;
    if (VEC4) {  // per_clip % 4 == 0: every clip base is 16-byte aligned
        const int n4 = per_clip >> 2;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
            float4 v = reinterpret_cast<float4*>(o)[i];
            v.x = (fmaxf(v.x, floor_v) + 4.0f) * 0.25f;
            v.y = (fmaxf(v.y, floor_v) + 4.0f) * 0.25f;
            v.z = (fmaxf(v.z, floor_v) + 4.0f) * 0.25f;
            v.w = (fmaxf(v.w, floor_v) + 4.0f) * 0.25f;
            reinterpret_cast<float4*>(o)[i] = v;
        }
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per_clip; i += gridDim.x * blockDim.x)
            o[i] = (fmaxf(o[i], floor_v) + 4.0f) * 0.25f;
    }
}

constexpr size_t TILE_LDS_BYTES = (size_t)TILE_S * 4 + (size_t)NGROUPS * 20 * EROW * 8 + (size_t)TILE_F * PSTRIDE * 4 +
                                  (size_t)(MAX_MELS + MAX_MELS + 1) * 4 + (size_t)MAX_NNZ * 4 + 8 * 4;
static_assert((size_t)NGROUPS * YGROUP * 8 <= (size_t)NGROUPS * 20 * EROW * 8 + (size_t)TILE_F * PSTRIDE * 4,
              "Y image must fit inside the E + P region it aliases");

}  // namespace

// Scratch of asrk_logmel_f32_noalloc: the per-clip maxima (ordered-integer keys), one word per clip.  In host code this
// name is only a stand-in for the symbol, not a device address: every device has its own copy at its own address, which
// asrk_logmel_plan_create looks up once (hipGetSymbolAddress, with the plan's device current) and keeps in the plan;
// kernels and hipMemsetAsync only ever get that pointer.
static __device__ __align__(256) unsigned g_noalloc_clip_max[ASRK_LOGMEL_NOALLOC_MAX_CLIPS];

struct asrk_logmel_plan {
    int n_mels;
    int device;
    MelTables* tables;          // device
    unsigned* noalloc_scratch;  // this device's address of g_noalloc_clip_max
};

extern "C" int asrk_logmel_plan_create(int n_mels, const float* mel_filters, asrk_logmel_plan** plan) {
    if (!mel_filters || !plan) return ASRK_ERR_NULL;
    if (n_mels < 1 || n_mels > MAX_MELS) return ASRK_ERR_SHAPE;
    MelTables* h = new MelTables();
    const double two_pi = 6.283185307179586476925286766559;
    for (int n = 0; n < NFFT; ++n) h->window[n] = (float)(0.5 - 0.5 * cos(two_pi * n / NFFT));
    for (int k1 = 0; k1 < 20; ++k1)
        for (int n2 = 0; n2 < 20; ++n2) {
            const double a = -two_pi * (double)(n2 * k1) / NFFT;
            h->twiddle[k1 * 20 + n2] = make_float2((float)cos(a), (float)sin(a));
        }
    int nnz = 0;
    for (int m = 0; m < n_mels; ++m) {
        int lo = NBINS, hi = -1;
        for (int k = 0; k < NBINS; ++k)
            if (mel_filters[(size_t)k * n_mels + m] != 0.0f) { lo = k < lo ? k : lo; hi = k; }
        h->mel_ptr[m] = nnz;
        h->mel_start[m] = hi < 0 ? 0 : lo;
        if (hi >= 0) {
            if (nnz + (hi - lo + 1) > MAX_NNZ) { delete h; return ASRK_ERR_SHAPE; }
            for (int k = lo; k <= hi; ++k) h->mel_w[nnz++] = mel_filters[(size_t)k * n_mels + m];
        }
    }
    h->mel_ptr[n_mels] = nnz;
    asrk_logmel_plan* p = new asrk_logmel_plan();
    p->n_mels = n_mels;
    if (hipGetDevice(&p->device) != hipSuccess || hipMalloc(&p->tables, sizeof(MelTables)) != hipSuccess ||
        hipMemcpy(p->tables, h, sizeof(MelTables), hipMemcpyHostToDevice) != hipSuccess) {
        delete h;
        delete p;
        return ASRK_ERR_HIP;
    }
    delete h;
    // one-time set-up for the entry point that allocates nothing: this device's address of the static scratch array, and
    // the tile kernel's LDS opt-in (so that the first asrk_logmel_f32_noalloc call changes no function attribute either)
    void* sym = nullptr;
    static std::atomic<uint64_t> lds_set{0};
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_noalloc_clip_max)) != hipSuccess || !sym ||
        asrk_max_lds_once((const void*)logmel_tile_kernel, (int)TILE_LDS_BYTES, lds_set) != ASRK_OK) {
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
int logmel_launch(const asrk_logmel_plan* plan, const float* wave, int B, int n_samples, float* out, unsigned* clip_max_key,
                  bool prof_ok, hipStream_t st) {
    const int n_frames = n_samples / HOP;
    static std::atomic<uint64_t> lds_set{0};
    if (asrk_max_lds_once((const void*)logmel_tile_kernel, (int)TILE_LDS_BYTES, lds_set) != ASRK_OK) return ASRK_ERR_HIP;
    ASRK_CHECK_HIP(hipMemsetAsync(clip_max_key, 0, (size_t)B * 4, st));
    // algorithmic bytes: read the waveform once, write the features once
    AsrkProfScope prof(prof_ok ? ASRK_PROF_LOGMEL : -1, (double)B * ((double)n_samples * 4 + (double)plan->n_mels * n_frames * 4), st);
    dim3 grid((n_frames + TILE_F - 1) / TILE_F, B);
    hipLaunchKernelGGL(logmel_tile_kernel, grid, dim3(NTHREADS), TILE_LDS_BYTES, st, wave, n_samples, n_frames,
                       plan->n_mels, plan->tables, out, clip_max_key);
    ASRK_CHECK_HIP(hipGetLastError());
    const int per_clip = plan->n_mels * n_frames;
    if ((per_clip & 3) == 0) {
        int gx = (per_clip / 4 + 255) / 256;
        gx = gx > 64 ? 64 : gx;
        hipLaunchKernelGGL(logmel_finish_kernel<true>, dim3(gx, B), dim3(256), 0, st, out, clip_max_key, per_clip);
    } else {
        int gx = (per_clip + 255) / 256;
        gx = gx > 64 ? 64 : gx;
        hipLaunchKernelGGL(logmel_finish_kernel<false>, dim3(gx, B), dim3(256), 0, st, out, clip_max_key, per_clip);
    }
    ASRK_CHECK_HIP(hipGetLastError());
    return ASRK_OK;
}

int logmel_check(const asrk_logmel_plan* plan, const float* wave, int B, int n_samples, const float* out) {
    if (!plan || !wave || !out) return ASRK_ERR_NULL;
    if (B < 0 || n_samples <= NFFT / 2) return ASRK_ERR_SHAPE;
    if (((uintptr_t)out & 15) || ((uintptr_t)wave & 3)) return ASRK_ERR_ALIGN;
    if (n_samples / HOP < 1 || B > 65535) return ASRK_ERR_SHAPE;
    return ASRK_OK;
}
}  // namespace

extern "C" int asrk_logmel_f32(const asrk_logmel_plan* plan, const float* wave, int B, int n_samples, float* out,
                               float* clip_max_ws, void* stream) {
    if (!clip_max_ws) return ASRK_ERR_NULL;
    const int rc = logmel_check(plan, wave, B, n_samples, out);
    if (rc != ASRK_OK) return rc;
    if (B == 0) return ASRK_OK;
    return logmel_launch(plan, wave, B, n_samples, out, reinterpret_cast<unsigned*>(clip_max_ws), true, (hipStream_t)stream);
}

// The variant that allocates nothing (include/asrk.h): the per-clip maxima live in this device's copy of
// g_noalloc_clip_max instead of a caller-provided workspace.  No hipMalloc / hipFree / event creation on this path.
extern "C" int asrk_logmel_f32_noalloc(const asrk_logmel_plan* plan, const float* wave, int B, int n_samples, float* out,
                                       void* stream) {
    const int rc = logmel_check(plan, wave, B, n_samples, out);
    if (rc != ASRK_OK) return rc;
    if (B > ASRK_LOGMEL_NOALLOC_MAX_CLIPS) return ASRK_ERR_WORKSPACE;  // beyond the static scratch: use asrk_logmel_f32
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return ASRK_ERR_HIP;
    if (dev != plan->device || !plan->noalloc_scratch) return ASRK_ERR_UNSUPPORTED;  // the scratch address is per device
    if (B == 0) return ASRK_OK;
    return logmel_launch(plan, wave, B, n_samples, out, plan->noalloc_scratch, false, (hipStream_t)stream);
}
