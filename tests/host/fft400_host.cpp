// Host emulation of the log-mel kernel's paired-frame 400-point FFT (asr_amd/csrc/asrk_fft400.h):
// walks the 16 x 20 lanes of one workgroup in loops, with the LDS images as plain arrays.
// usage: fft400_host <in: 5360 f32 samples> <out: 32*201 f32 power>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../asr_amd/csrc/asrk_fft400.h"

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    std::vector<float> S(TILE_S);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(S.data(), 4, TILE_S, f) != (size_t)TILE_S) return 3;
    fclose(f);
    MelTables* tab = new MelTables();
    const double two_pi = 6.283185307179586476925286766559;
    for (int n = 0; n < NFFT; ++n) tab->window[n] = (float)(0.5 - 0.5 * cos(two_pi * n / NFFT));
    for (int k1 = 0; k1 < 20; ++k1)
        for (int n2 = 0; n2 < 20; ++n2) {
            const double a = -two_pi * (double)(n2 * k1) / NFFT;
            tab->twiddle[k1 * 20 + n2].x = (float)cos(a);
            tab->twiddle[k1 * 20 + n2].y = (float)sin(a);
        }
    std::vector<float2> Y(NGROUPS * YGROUP), E(NGROUPS * 20 * EROW);
    std::vector<float> P(TILE_F * PSTRIDE, -1.0f);
    static float zr[NGROUPS][20][20], zi[NGROUPS][20][20];
    for (int g = 0; g < NGROUPS; ++g)
        for (int j = 0; j < 20; ++j) fft400_stage1(S.data(), Y.data(), tab, g, j);
    for (int g = 0; g < NGROUPS; ++g)
        for (int j = 0; j < 20; ++j) fft400_stage2(Y.data(), g, j, zr[g][j], zi[g][j]);
    for (int g = 0; g < NGROUPS; ++g)
        for (int j = 0; j < 20; ++j) fft400_publish(E.data(), g, j, zr[g][j], zi[g][j]);
    for (int g = 0; g < NGROUPS; ++g)
        for (int j = 0; j < 20; ++j) fft400_power(E.data(), P.data(), g, j, zr[g][j], zi[g][j]);
    f = fopen(argv[2], "wb");
    fwrite(P.data(), 4, P.size(), f);
    fclose(f);
    return 0;
}
