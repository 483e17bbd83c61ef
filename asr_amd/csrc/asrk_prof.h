// Optional per-launch event timing (off by default; zero cost when off beyond one branch).
#pragma once
#include <hip/hip_runtime.h>

// One category per rocprof symbol of the hot kernels: the three GEMM kernels are told apart so that bench.py's roofline
// names the kernel that actually dominated (0 = gemm256p_kernel, 5 = gemm256_kernel, 6 = gemm_bf16_kernel).
enum { ASRK_PROF_GEMM = 0, ASRK_PROF_ATTENTION = 1, ASRK_PROF_LOGMEL = 2, ASRK_PROF_NORM = 3, ASRK_PROF_SPLICE = 4,
       ASRK_PROF_GEMM256 = 5, ASRK_PROF_GEMM128 = 6 };

bool asrk_prof_enabled();
int asrk_prof_begin(int cat, double work, hipStream_t st);
void asrk_prof_end(int id, hipStream_t st);

struct AsrkProfScope {
    int id;
    hipStream_t st;
    AsrkProfScope(int cat, double work, hipStream_t s) : id(-1), st(s) {
        if (cat >= 0 && asrk_prof_enabled()) id = asrk_prof_begin(cat, work, s);  // cat < 0: never profiled
    }
    ~AsrkProfScope() {
        if (id >= 0) asrk_prof_end(id, st);
    }
};
