// The radix-8 passes of the 512-point transform that carry the register part of the negacyclic twist (round 4), as plain
// arithmetic on a two-component value type: compiled into the kernels through fft512.h (V = double2) and, on the host, into
// tests/native/dft8_twist_test.cpp, which checks them against the defining sums.
#pragma once

#include <cmath>

#ifndef IEACHE_HD
#ifdef __HIPCC__
#define IEACHE_HD __device__ __forceinline__
#else
#define IEACHE_HD inline
#endif
#endif

namespace ieache {
namespace w64 {

constexpr double kR = 0.70710678118654752440;  // 1/sqrt(2)

template <class V>
IEACHE_HD V mk(double a, double b) {
    V v;
    v.x = a;
    v.y = b;
    return v;
}

// ---- radix-8 passes with the register part of the negacyclic twist folded in (round 4) ----
// The first forward pass runs over r, the register index of coefficient j = 64 r + lane, whose twist factor is e^{i pi r/16}:
// X_k = sum_r y_r e^{i pi r/16} W8^{rk} is the evaluation of sum_r y_r z^r at the eight 8th roots of i, i.e. three radix-2
// decimation-in-time stages with ONE twiddle per stage and half (z^8 - i = (z^4 - w)(z^4 + w), w = e^{i pi/4}, ...).  With each
// twiddle written as cos (1 + i tan) a butterfly a +- w b is two FMAs for the rotation and four for the sums: 72 FP64
// instructions where seven twist multiplies (28) + dft8 (52) take 80.  Every |tan| <= 0.67.
constexpr double kTan8 = 0.41421356237309504880, kCos8 = 0.92387953251128675613;     // pi/8
constexpr double kTan16 = 0.19891236737965800691, kCos16 = 0.98078528040323044913;   // pi/16
constexpr double kTan316 = 0.66817863791929891999, kCos316 = 0.83146961230254523708; // 3 pi/16
template <class V>
IEACHE_HD void dft8_twist_fwd(V (&x)[8]) {
    V Ap[4], Am[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {  // a +- e^{i pi/4} b
        const double ux = x[j + 4].x - x[j + 4].y, uy = x[j + 4].x + x[j + 4].y;
        Ap[j] = mk<V>(fma(kR, ux, x[j].x), fma(kR, uy, x[j].y));
        Am[j] = mk<V>(fma(-kR, ux, x[j].x), fma(-kR, uy, x[j].y));
    }
    V Bpp[2], Bpm[2], Bmp[2], Bmm[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {  // a +- e^{i pi/8} b  (outputs k = 0,4 | 2,6) ;  a +- (-i) e^{i pi/8} b  (k = 1,5 | 3,7)
        double ux = fma(-kTan8, Ap[j + 2].y, Ap[j + 2].x), uy = fma(kTan8, Ap[j + 2].x, Ap[j + 2].y);
        Bpp[j] = mk<V>(fma(kCos8, ux, Ap[j].x), fma(kCos8, uy, Ap[j].y));
        Bpm[j] = mk<V>(fma(-kCos8, ux, Ap[j].x), fma(-kCos8, uy, Ap[j].y));
        ux = fma(-kTan8, Am[j + 2].y, Am[j + 2].x), uy = fma(kTan8, Am[j + 2].x, Am[j + 2].y);
        Bmp[j] = mk<V>(fma(kCos8, uy, Am[j].x), fma(-kCos8, ux, Am[j].y));
        Bmm[j] = mk<V>(fma(-kCos8, uy, Am[j].x), fma(kCos8, ux, Am[j].y));
    }
    // z = e^{i pi/16} W8^k0:  k0 = 0: e^{i pi/16};  2: (-i) e^{i pi/16};  1: e^{-3 i pi/16};  3: (-i) e^{-3 i pi/16}
    {
        const double ux = fma(-kTan16, Bpp[1].y, Bpp[1].x), uy = fma(kTan16, Bpp[1].x, Bpp[1].y);
        x[0] = mk<V>(fma(kCos16, ux, Bpp[0].x), fma(kCos16, uy, Bpp[0].y));
        x[4] = mk<V>(fma(-kCos16, ux, Bpp[0].x), fma(-kCos16, uy, Bpp[0].y));
    }
    {
        const double ux = fma(-kTan16, Bpm[1].y, Bpm[1].x), uy = fma(kTan16, Bpm[1].x, Bpm[1].y);
        x[2] = mk<V>(fma(kCos16, uy, Bpm[0].x), fma(-kCos16, ux, Bpm[0].y));
        x[6] = mk<V>(fma(-kCos16, uy, Bpm[0].x), fma(kCos16, ux, Bpm[0].y));
    }
    {
        const double ux = fma(kTan316, Bmp[1].y, Bmp[1].x), uy = fma(-kTan316, Bmp[1].x, Bmp[1].y);
        x[1] = mk<V>(fma(kCos316, ux, Bmp[0].x), fma(kCos316, uy, Bmp[0].y));
        x[5] = mk<V>(fma(-kCos316, ux, Bmp[0].x), fma(-kCos316, uy, Bmp[0].y));
    }
    {
        const double ux = fma(kTan316, Bmm[1].y, Bmm[1].x), uy = fma(-kTan316, Bmm[1].x, Bmm[1].y);
        x[3] = mk<V>(fma(kCos316, uy, Bmm[0].x), fma(-kCos316, ux, Bmm[0].y));
        x[7] = mk<V>(fma(-kCos316, uy, Bmm[0].x), fma(kCos316, ux, Bmm[0].y));
    }
}
// The last inverse pass is the conjugate transpose of that flow graph (decimation in frequency: sum, and the difference
// rotated by the conjugate twiddle).  The cosines are left pending as REAL scales -- equal within every pair the next stage
// combines, but for one ratio -- and come out as one factor per output register, which the caller folds, with the 1/512 of
// the transform, into the FMA that rounds: true y_r e^{-i pi r/16} / 512 = x[r] * untwist_gain(r).  72 instructions where
// dft8 (52) + seven untwist multiplies (28) take 80, and the rounding add becomes an FMA at no cost.
IEACHE_HD constexpr double untwist_gain(int r) {
    constexpr double g[8] = {1.0, kCos16, kCos8, kCos8 * kCos16, kR, kR * kCos16, kR * kCos8, kR * kCos8 * kCos16};
    return g[r] * (1.0 / 512.0);
}
template <class V>
IEACHE_HD void dft8_untwist_inv(V (&x)[8]) {
    V Bpp[2], Bpm[2], Bmp[2], Bmm[2];
    {   // k0 = 0: e^{-i pi/16} (X_0 - X_4), cos pending
        const double dx = x[0].x - x[4].x, dy = x[0].y - x[4].y;
        Bpp[0] = mk<V>(x[0].x + x[4].x, x[0].y + x[4].y);
        Bpp[1] = mk<V>(fma(kTan16, dy, dx), fma(-kTan16, dx, dy));
    }
    {   // k0 = 2: i e^{-i pi/16} (X_2 - X_6)
        const double dx = x[2].x - x[6].x, dy = x[2].y - x[6].y;
        Bpm[0] = mk<V>(x[2].x + x[6].x, x[2].y + x[6].y);
        Bpm[1] = mk<V>(-fma(-kTan16, dx, dy), fma(kTan16, dy, dx));
    }
    {   // k0 = 1: e^{+3 i pi/16} (X_1 - X_5)
        const double dx = x[1].x - x[5].x, dy = x[1].y - x[5].y;
        Bmp[0] = mk<V>(x[1].x + x[5].x, x[1].y + x[5].y);
        Bmp[1] = mk<V>(fma(-kTan316, dy, dx), fma(kTan316, dx, dy));
    }
    {   // k0 = 3: i e^{+3 i pi/16} (X_3 - X_7)
        const double dx = x[3].x - x[7].x, dy = x[3].y - x[7].y;
        Bmm[0] = mk<V>(x[3].x + x[7].x, x[3].y + x[7].y);
        Bmm[1] = mk<V>(-fma(kTan316, dx, dy), fma(-kTan316, dy, dx));
    }
    V Ap[4], Am[4];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        double dx = Bpp[j].x - Bpm[j].x, dy = Bpp[j].y - Bpm[j].y;
        Ap[j] = mk<V>(Bpp[j].x + Bpm[j].x, Bpp[j].y + Bpm[j].y);
        Ap[j + 2] = mk<V>(fma(kTan8, dy, dx), fma(-kTan8, dx, dy));     // e^{-i pi/8} d, cos pending
        dx = Bmp[j].x - Bmm[j].x, dy = Bmp[j].y - Bmm[j].y;
        Am[j] = mk<V>(Bmp[j].x + Bmm[j].x, Bmp[j].y + Bmm[j].y);
        Am[j + 2] = mk<V>(-fma(-kTan8, dx, dy), fma(kTan8, dy, dx));    // i e^{-i pi/8} d
    }
    // pending: Ap = {1, c16, c8, c8 c16}, Am = {1, c316, c8, c8 c316}: the odd pairs meet through the ratio c316 / c16
    constexpr double rho = kCos316 / kCos16;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double sx, sy, dx, dy;
        if (j & 1) {
            sx = fma(rho, Am[j].x, Ap[j].x), sy = fma(rho, Am[j].y, Ap[j].y);
            dx = fma(-rho, Am[j].x, Ap[j].x), dy = fma(-rho, Am[j].y, Ap[j].y);
        } else {
            sx = Ap[j].x + Am[j].x, sy = Ap[j].y + Am[j].y;
            dx = Ap[j].x - Am[j].x, dy = Ap[j].y - Am[j].y;
        }
        x[j] = mk<V>(sx, sy);
        x[j + 4] = mk<V>(dx + dy, dy - dx);  // e^{-i pi/4} d, 1/sqrt2 pending
    }
}

}  // namespace w64
}  // namespace ieache
