// Host emulation of the log-mel kernel's paired-frame 400-point FFT (asr_amd/csrc/asrk_fft400.h): walks the 64 lanes of
// ONE wave (6 frames = 3 frame pairs of 20 lanes, lanes 60..63 shadowing lanes 40..43) through the same per-lane steps
// as the device code, with the wave-private LDS image as a plain array and the cross-lane mirror fetch as an array
// lookup.  usage: fft400_host <in: 1200 f32 samples> <out: 6*201 f32 power>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../asr_amd/csrc/asrk_fft400.h"


int main(int argc, char** argv) {
    if (argc != 3) return 2;
    std::vector<float> S(WAVE_S);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(S.data(), 4, WAVE_S, f) != (size_t)WAVE_S) return 3;
    fclose(f);
    MelTables* tab = new MelTables();
    const double two_pi = 6.283185307179586476925286766559;
    for (int n = 0; n < NFFT; ++n) tab->window[n] = (float)(0.5 * (0.5 - 0.5 * cos(two_pi * n / NFFT)));  // Hann / 2
    for (int k1 = 0; k1 < 20; ++k1)
        for (int n2 = 0; n2 < 20; ++n2) {
            const double a = -two_pi * (double)(n2 * k1) / NFFT;
            tab->twiddle[k1 * 20 + n2].x = (float)cos(a);
            tab->twiddle[k1 * 20 + n2].y = (float)sin(a);
        }
    std::vector<float> priv(PRIV_FLOATS, -7.0f);
    static float re[64][20], im[64][20], zr[64][20], zi[64][20];
    // stage 1, every lane (the shadow lanes 60..63 repeat lanes 40..43)
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane_pair(lane), j = lane_j(lane);
        for (int n1 = 0; n1 < 20; ++n1) {
            re[lane][n1] = S[(2 * g) * HOP + 20 * n1 + j];
            im[lane][n1] = S[(2 * g + 1) * HOP + 20 * n1 + j];
        }
        float win[20];
        for (int n1 = 0; n1 < 20; ++n1) win[n1] = tab->window[20 * n1 + j];
        fft400_stage1(re[lane], im[lane], win, tab->twiddle, j);
    }
    // exchange: all lanes write, then all lanes read their row -- first the real parts, then the imaginary parts
    for (int pass = 0; pass < 2; ++pass) {
        for (int lane = 0; lane < 64; ++lane)
            for (int k1 = 0; k1 < 20; ++k1)
                priv[xch_index(lane_pair(lane), k1, lane_j(lane))] = pass ? im[lane][k1] : re[lane][k1];
        for (int lane = 0; lane < 64; ++lane)
            for (int n2 = 0; n2 < 20; ++n2)
                (pass ? zi : zr)[lane][n2] = priv[xch_index(lane_pair(lane), lane_j(lane), n2)];
    }
    for (int lane = 0; lane < 64; ++lane) dft20(zr[lane], zi[lane]);
    // mirror fetch + power image [frame][bin] in the same private region
    std::vector<float> P(FPW * PBINS, -1.0f);
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane_pair(lane), j = lane_j(lane), ml = mirror_lane(g, j);
        for (int k2 = 0; k2 <= 10; ++k2) {
            if (k2 == 10 && j != 0) continue;
            float cr, ci;
            if (j == 0) { cr = zr[lane][mirror_reg_lane0(k2)]; ci = zi[lane][mirror_reg_lane0(k2)]; }
            else { cr = zr[ml][mirror_reg(k2)]; ci = zi[ml][mirror_reg(k2)]; }
            fft400_power(zr[lane][k2], zi[lane][k2], cr, ci, P[(2 * g) * PBINS + j + 20 * k2],
                         P[(2 * g + 1) * PBINS + j + 20 * k2]);
        }
    }
    f = fopen(argv[2], "wb");
    fwrite(P.data(), 4, P.size(), f);
    fclose(f);
    return 0;
}
