// On-device collate: the padding / truncation / mask step of the feature extractor
// (HF feature_extraction_whisper.py:300-307, 332-341; sequence_utils.py:224-337: right-pad with 0.0 or truncate to
// `target` samples, sample-level mask taken every hop) done on the GPU from ONE ragged buffer, so that the host ships
// only the valid samples -- as int16 PCM (half the bytes again) or float32 -- in a single H2D copy instead of padding
// every clip to 30 s on the CPU.  int16 is scaled by 1/32768 (the int16 -> float convention of libsndfile, which is
// what `datasets` decodes the reference's audio with); the division is exact in fp32.
#include "asrk_common.h"

namespace {

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<short>(short v) { return (float)v * (1.0f / 32768.0f); }

// pcm: ragged samples of B clips back to back, clip b = pcm[offsets[b] .. offsets[b+1]).  wave [B][target] f32,
// mask [B][target / hop] int32 (1 where frame * hop < min(len, target)), either output optional.
template <typename T>
__global__ __launch_bounds__(256) void collate_kernel(const T* __restrict__ pcm, const long long* __restrict__ offsets,
                                                      int target, int hop, float* __restrict__ wave, int* __restrict__ mask) {
    const int b = blockIdx.y;
    const long long o0 = offsets[b];
    long long len = offsets[b + 1] - o0;
    len = len < 0 ? 0 : (len > target ? target : len);
    const T* src = pcm + o0;
    const int n_mask = target / hop;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < target; i += gridDim.x * blockDim.x) {
        if (wave) wave[(size_t)b * target + i] = i < len ? to_f32<T>(src[i]) : 0.0f;
        if (mask && i < n_mask) mask[(size_t)b * n_mask + i] = (long long)i * hop < len ? 1 : 0;
    }
}

}  // namespace

extern "C" int asrk_collate_pcm(const void* pcm, int pcm_dtype, const int64_t* offsets, int B, int target, int hop, float* wave,
                                int* frame_mask, void* stream) {
    if (!pcm || !offsets || (!wave && !frame_mask)) return ASRK_ERR_NULL;
    if (B <= 0 || B > 65535 || target <= 0 || hop <= 0) return ASRK_ERR_SHAPE;
    const dim3 grid((unsigned)((target + 256 * 8 - 1) / (256 * 8)), (unsigned)B);
    if (pcm_dtype == ASRK_F32)
        hipLaunchKernelGGL(collate_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)pcm,
                           (const long long*)offsets, target, hop, wave, frame_mask);
    else if (pcm_dtype == ASRK_I16)
        hipLaunchKernelGGL(collate_kernel<short>, grid, dim3(256), 0, (hipStream_t)stream, (const short*)pcm,
                           (const long long*)offsets, target, hop, wave, frame_mask);
    else
        return ASRK_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}
