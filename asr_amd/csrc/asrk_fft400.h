// Paired-frame 400-point FFT pieces of the log-mel kernel, written so that the same source compiles for the device
// (asrk_logmel.hip) and for a host emulation (tests/host/fft400_host.cpp) that walks the lanes of one wave in a loop to
// check the index algebra without a GPU.
//
// Geometry.  A wave (64 lanes) owns 6 consecutive frames = 3 frame pairs; pair g is transformed by the 20 lanes
// 20 g .. 20 g + 19 (lanes 60..63 shadow lanes 40..43: same addresses, same values) as ONE complex 400-point FFT of
// z = x_even_frame + i x_odd_frame, factored 400 = 20 x 20:
//   stage 1  lane j: 20-point DFT over n1 of z[20 n1 + j] (registers), times W400^(k1 j)
//   exchange a 20 x 20 transpose inside the pair, through a wave-private LDS image [pair][k1][j] of plain floats
//            (real parts, then imaginary parts through the same image).  It is written with ds_write_b32 and read back
//            by rows with ds_read_b128; the dense 80-byte rows and the 1600-byte pair stride are conflict-free for both
//            (bank arithmetic in DESIGN.md 4.1).  A wave's LDS operations execute in order, so no barrier is involved.
//   stage 2  lane k1: 20-point DFT over n2 -> Z[k1 + 20 k2] in register k2
//   mirror   X_even[k] = (Z[k] + conj Z[400-k]) / 2, X_odd[k] = (Z[k] - conj Z[400-k]) / 2i: bin 400 - k lives in lane
//            (20 - k1) % 20 of the same pair, register 19 - k2 (lane 0: its own register (20 - k2) % 20), fetched by a
//            cross-lane ds_bpermute -- no LDS image, no barrier.
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
constexpr int FPW = 6;                               // frames per wave
constexpr int PAIRS = FPW / 2;                       // frame pairs per wave
constexpr int NWAVES = 4;                            // waves per workgroup
constexpr int TILE_F = FPW * NWAVES;                 // 24 frames per workgroup tile (3000 = 125 tiles)
constexpr int NTHREADS = 64 * NWAVES;                // 256
constexpr int WAVE_S = (FPW - 1) * HOP + NFFT;       // 1200 samples under a wave's 6 frames
constexpr int XPAIR = 400;                           // floats per pair of the exchange image [k1][j], dense
constexpr int PBINS = 204;                           // bins per frame of the power image: 0..200 + 3 zeros (whole 4-bin chunks)
constexpr int PPAIR = 424;                           // floats per pair of the power image [bin][2 frames]: 408 + 16, so that the
                                                     // 8-byte writes of the pairs in one half-wave fall on disjoint banks
constexpr int PRIV_FLOATS = PAIRS * PPAIR;           // wave-private LDS: max(exchange image 3 * 400, power image 3 * 424)
constexpr int MAX_MELS = 128;
constexpr int MAX_NNZ = 12 * 20 * 4;                 // weights [step][list][4]
static_assert(PRIV_FLOATS >= PAIRS * XPAIR && PRIV_FLOATS % 4 == 0 && PPAIR >= 2 * PBINS && PPAIR % 4 == 0, "private region");

constexpr int MEL_LANES = 20;                        // lanes per frame PAIR in the mel projection (3 pairs x 20 = 60 lanes)
constexpr int MAX_CHUNKS = 12;                       // chunks per mel lane, at most (10 / 10 for the slaney banks)

// The slaney bank by rows, sparse, cut into 16-byte chunks on the bin grid: filter m covers the bins lo..hi, i.e. the
// chunks lo / 4 .. hi / 4 of 4 bins each, with zero weights outside lo..hi.  The power image holds the two frames of a
// pair side by side, [bin][2], so a mel lane = (frame pair, list q of filters) multiplies one weight chunk into both
// frames: per chunk one descriptor, one ds_read_b128 of 4 weights and two of 4 bins x 2 frames, 4 packed FMAs.  The
// filters are dealt to the 20 lists so that the lists are equally long (longest first onto the shortest list: 9 / 10
// chunks for the 80 / 128-filter slaney banks against 161 / 196 chunks in all), padded with all-zero chunks to one
// wave-uniform, even trip count:
// mel_chunk[e * 20 + q] = chunk of bins | filter << 14 | (last chunk of its filter) << 21; its four weights are
// mel_w[e][q][0..3] (for odd q with the halves swapped: the kernel reads the two halves of a power chunk in opposite order
// there, to spread the banks).
constexpr int CHUNK_LAST = 1 << 21;
struct MelTables {      // device-resident, built once per plan
    float window[NFFT];          // periodic Hann / 2: the |X|^2 of the pair separation carries a factor 1/4
    float2 twiddle[NFFT];        // W400^(j*k1) stored at [k1*20 + j]
    int mel_chunk[MAX_CHUNKS * MEL_LANES];
    float mel_w[MAX_NNZ];        // [step][list][4]
};

// lane -> (pair, index in pair); lanes 60..63 shadow lanes 40..43
ASRK_HD int lane_pair(int lane) { return lane < 60 ? lane / 20 : 2; }
ASRK_HD int lane_j(int lane) { return lane < 60 ? lane % 20 : lane - 60; }
// exchange image: element (k1, j) of pair g
ASRK_HD int xch_index(int g, int k1, int j) { return g * XPAIR + k1 * 20 + j; }
// mirror: the lane that holds bins congruent to -k1, and the register of bin 400 - (k1 + 20 k2) there
ASRK_HD int mirror_lane(int g, int j) { return g * 20 + (j ? 20 - j : 0); }
ASRK_HD constexpr int mirror_reg(int k2) { return 19 - k2; }              // lanes j > 0
ASRK_HD constexpr int mirror_reg_lane0(int k2) { return (20 - k2) % 20; }  // lane j == 0 (its own register)

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

// stage 1 of lane j: re / im hold the raw samples of the even / odd frame at n = 20 n1 + j on entry, the twiddled
// column Y[k1][j] on exit.  `win` = the lane's 20 window values window[20 n1 + j] (registers: the kernel is persistent);
// `tw` = the twiddle table (LDS on the device).
ASRK_HD void fft400_stage1(float (&re)[20], float (&im)[20], const float (&win)[20], const float2* tw, int j) {
#pragma unroll
    for (int n1 = 0; n1 < 20; ++n1) {
        re[n1] *= win[n1];
        im[n1] *= win[n1];
    }
    dft20(re, im);
#pragma unroll
    for (int k1 = 1; k1 < 20; ++k1) {  // k1 = 0: W^0 = 1
        const float2 t = tw[k1 * 20 + j];
        const float r = re[k1] * t.x - im[k1] * t.y;
        im[k1] = re[k1] * t.y + im[k1] * t.x;
        re[k1] = r;
    }
}

// power of bins k = j + 20 k2 of the even frame (p0) and the odd frame (p1) from Z[k] = (zr, zi)[k2] and the mirror
// value c = Z[400 - k]
ASRK_HD void fft400_power(float zr, float zi, float cr, float ci, float& p0, float& p1) {
    const float ar = zr + cr, ai = zi - ci;  // 2 * X_even[k]
    const float br = zr - cr, bi = zi + ci;  // 2i * X_odd[k]
    p0 = ar * ar + ai * ai;  // the factor 1/4 is in the window (Hann / 2)
    p1 = br * br + bi * bi;
}
