// Paired-frame 400-point FFT pieces of the log-mel kernel, written so that the same source
// compiles for the device (asrk_logmel.hip) and for a host emulation (tests/host/fft400_host.cpp)
// that walks the lanes in a loop to check the index algebra without a GPU.
#pragma once
#ifdef __HIPCC__
#define ASRK_HD __device__ __forceinline__
#else
#define ASRK_HD static inline
struct float2 { float x, y; };
#endif

constexpr int NFFT = 400;
constexpr int HOP = 160;
constexpr int NBINS = 201;
constexpr int TILE_F = 32;                          // frames per workgroup
constexpr int NGROUPS = 16;                         // frame pairs per workgroup
constexpr int NTHREADS = NGROUPS * 20;              // 320
constexpr int TILE_S = (TILE_F - 1) * HOP + NFFT;   // 5360 samples
constexpr int YROW = 21;                            // complex row stride of the 20x20 exchange (conflict-free)
constexpr int YGROUP = 20 * YROW;                   // 420 complex per group
constexpr int EROW = 11;                            // complex per lane in the mirror exchange
constexpr int PSTRIDE = 201;                        // floats per frame of the power image (odd: conflict-free)
constexpr int MAX_MELS = 128;
constexpr int MAX_NNZ = 512;

struct MelTables {      // device-resident, built once per plan
    float window[NFFT];          // periodic Hann
    float2 twiddle[NFFT];        // W400^(n2*k1) stored at [k1*20 + n2]
    int mel_start[MAX_MELS];     // first FFT bin of each filter
    int mel_ptr[MAX_MELS + 1];   // CSR offsets into mel_w
    float mel_w[MAX_NNZ];
};

// ---- 20-point complex DFT in registers: 4 x 5 Good-Thomas (input n = (5 n1 + 4 n2) % 20,
//      output k = (5 k1 + 16 k2) % 20), forward sign e^{-i...}.
ASRK_HD void dft20(float (&re)[20], float (&im)[20]) {
    float tr[20], ti[20];  // tr[k1*5 + n2]
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        const int i0 = (4 * n2) % 20, i1 = (5 + 4 * n2) % 20, i2 = (10 + 4 * n2) % 20, i3 = (15 + 4 * n2) % 20;
        const float ar = re[i0] + re[i2], ai = im[i0] + im[i2];
        const float br = re[i0] - re[i2], bi = im[i0] - im[i2];
        const float cr = re[i1] + re[i3], ci = im[i1] + im[i3];
        const float dr = re[i1] - re[i3], di = im[i1] - im[i3];
        tr[0 * 5 + n2] = ar + cr;  ti[0 * 5 + n2] = ai + ci;
        tr[2 * 5 + n2] = ar - cr;  ti[2 * 5 + n2] = ai - ci;
        // k1 = 1: b + (-i) d ; k1 = 3: b + (i) d
        tr[1 * 5 + n2] = br + di;  ti[1 * 5 + n2] = bi - dr;
        tr[3 * 5 + n2] = br - di;  ti[3 * 5 + n2] = bi + dr;
    }
    constexpr float C1 = 0.30901699437494742f, C2 = -0.80901699437494742f;
    constexpr float S1 = 0.95105651629515357f, S2 = 0.58778525229247313f;
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        const float x0r = tr[k1 * 5 + 0], x0i = ti[k1 * 5 + 0];
        const float a1r = tr[k1 * 5 + 1] + tr[k1 * 5 + 4], a1i = ti[k1 * 5 + 1] + ti[k1 * 5 + 4];
        const float b1r = tr[k1 * 5 + 1] - tr[k1 * 5 + 4], b1i = ti[k1 * 5 + 1] - ti[k1 * 5 + 4];
        const float a2r = tr[k1 * 5 + 2] + tr[k1 * 5 + 3], a2i = ti[k1 * 5 + 2] + ti[k1 * 5 + 3];
        const float b2r = tr[k1 * 5 + 2] - tr[k1 * 5 + 3], b2i = ti[k1 * 5 + 2] - ti[k1 * 5 + 3];
        const float r1r = x0r + C1 * a1r + C2 * a2r, r1i = x0i + C1 * a1i + C2 * a2i;
        const float r2r = x0r + C2 * a1r + C1 * a2r, r2i = x0i + C2 * a1i + C1 * a2i;
        const float i1r = S1 * b1r + S2 * b2r, i1i = S1 * b1i + S2 * b2i;
        const float i2r = S2 * b1r - S1 * b2r, i2i = S2 * b1i - S1 * b2i;
        const int o0 = (5 * k1) % 20, o1 = (5 * k1 + 16) % 20, o2 = (5 * k1 + 32) % 20, o3 = (5 * k1 + 48) % 20,
                  o4 = (5 * k1 + 64) % 20;
        re[o0] = x0r + a1r + a2r;  im[o0] = x0i + a1i + a2i;
        re[o1] = r1r + i1i;        im[o1] = r1i - i1r;   // r1 - i*i1
        re[o4] = r1r - i1i;        im[o4] = r1i + i1r;   // r1 + i*i1
        re[o2] = r2r + i2i;        im[o2] = r2i - i2r;
        re[o3] = r2r - i2i;        im[o3] = r2i + i2r;
    }
}


// ---- the four per-lane steps of the paired-frame 400-point FFT.  (g, j) = (frame pair, lane in group).
// stage 1: z[n1] = w * (x_f0 + i x_f1) at n = 20 n1 + j; 20-pt DFT over n1; twiddle; Y[g][k1][j]
ASRK_HD void fft400_stage1(const float* S, float2* Y, const MelTables* tab, int g, int j) {
    float re[20], im[20];
    const float* f0 = S + (2 * g) * HOP;
    const float* f1 = f0 + HOP;
#pragma unroll
    for (int n1 = 0; n1 < 20; ++n1) {
        const float w = tab->window[20 * n1 + j];
        re[n1] = w * f0[20 * n1 + j];
        im[n1] = w * f1[20 * n1 + j];
    }
    dft20(re, im);
    float2* yrow = Y + g * YGROUP + j;
#pragma unroll
    for (int k1 = 0; k1 < 20; ++k1) {
        const float2 tw = tab->twiddle[k1 * 20 + j];
        float2 v;
        v.x = re[k1] * tw.x - im[k1] * tw.y;
        v.y = re[k1] * tw.y + im[k1] * tw.x;
        yrow[k1 * YROW] = v;
    }
}
// stage 2: lane (g, k1=j): 20-pt DFT over n2 -> Z[k1 + 20 k2] in register k2
ASRK_HD void fft400_stage2(const float2* Y, int g, int j, float (&zr)[20], float (&zi)[20]) {
    const float2* yrow = Y + g * YGROUP + j * YROW;
#pragma unroll
    for (int n2 = 0; n2 < 20; ++n2) {
        const float2 v = yrow[n2];
        zr[n2] = v.x;
        zi[n2] = v.y;
    }
    dft20(zr, zi);
}
// mirror exchange: publish Z[k1 + 20 k2'] for k2' = 10..19 (slots 0..9) and k2' = 0 (slot 10)
ASRK_HD void fft400_publish(float2* E, int g, int j, const float (&zr)[20], const float (&zi)[20]) {
    float2* e = E + (g * 20 + j) * EROW;
#pragma unroll
    for (int q = 0; q < 10; ++q) { float2 v; v.x = zr[10 + q]; v.y = zi[10 + q]; e[q] = v; }
    float2 v0; v0.x = zr[0]; v0.y = zi[0];
    e[10] = v0;
}
// separate the two real spectra and write |X|^2 for bins 0..200 of frames 2g and 2g+1
ASRK_HD void fft400_power(const float2* E, float* P, int g, int j, const float (&zr)[20], const float (&zi)[20]) {
    const int jp = (20 - j) % 20;  // lane holding bins congruent to -k1
    const float2* e = E + (g * 20 + jp) * EROW;
    float* p0 = P + (2 * g) * PSTRIDE + j;
    float* p1 = p0 + PSTRIDE;
#pragma unroll
    for (int k2 = 0; k2 <= 10; ++k2) {
        if (k2 < 10 || j == 0) {
            // partner bin 400 - k: slot 9 - k2 (k1 > 0) or 10 - k2 (k1 == 0; slot 10 = Z[0], slot 0 = Z[200])
            const float2 c = e[(j == 0) ? (10 - k2) : (9 - k2)];
            const float ar = zr[k2] + c.x, ai = zi[k2] - c.y;  // 2 * X1[k]
            const float br = zr[k2] - c.x, bi = zi[k2] + c.y;  // 2i * X2[k]
            p0[20 * k2] = 0.25f * (ar * ar + ai * ai);
            p1[20 * k2] = 0.25f * (br * br + bi * bi);
        }
    }
}
