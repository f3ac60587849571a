// Host check of the twisted radix-8 passes (ie-ache_amd/csrc/dft8_twist.h) against their defining sums:
//   forward  X_k = sum_r y_r e^{i pi r/16} e^{-2 pi i r k/8}
//   inverse  y_r = untwist_gain(r)-scaled:  x[r] * 512 * untwist_gain(r) = e^{-i pi r/16} sum_k X_k e^{+2 pi i r k/8}
// on digit-sized integer inputs (forward) and spectrum-sized values (inverse).  Prints the largest errors; exit 0 = within bounds.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>

#include "dft8_twist.h"

struct D2 {
    double x, y;
};

int main() {
    using cd = std::complex<double>;
    using namespace ieache::w64;
    const double PI = 3.14159265358979323846;
    double worst_f = 0, worst_i = 0;
    srand(12345);
    for (int trial = 0; trial < 4000; trial++) {
        D2 x[8];
        cd y[8];
        for (int r = 0; r < 8; r++) {
            y[r] = cd(rand() % 128 - 64, rand() % 128 - 64);  // the gadget digits: [-64, 64)
            x[r] = {y[r].real(), y[r].imag()};
        }
        dft8_twist_fwd(x);
        for (int k = 0; k < 8; k++) {
            cd X = 0;
            for (int r = 0; r < 8; r++) X += y[r] * std::polar(1.0, PI * r / 16) * std::polar(1.0, -2 * PI * r * k / 8);
            worst_f = fmax(worst_f, std::abs(X - cd(x[k].x, x[k].y)));
        }
        cd Xs[8];
        D2 z[8];
        const double mag = ldexp(1.0, 20 + trial % 24);  // up to ~2^44: what the last inverse pass sees
        for (int k = 0; k < 8; k++) {
            Xs[k] = cd((rand() % 2001 - 1000) * 1e-3 * mag, (rand() % 2001 - 1000) * 1e-3 * mag);
            z[k] = {Xs[k].real(), Xs[k].imag()};
        }
        dft8_untwist_inv(z);
        for (int r = 0; r < 8; r++) {
            cd Y = 0;
            for (int k = 0; k < 8; k++) Y += Xs[k] * std::polar(1.0, 2 * PI * r * k / 8);
            Y *= std::polar(1.0, -PI * r / 16);
            const double g = untwist_gain(r) * 512.0;
            worst_i = fmax(worst_i, std::abs(Y - cd(z[r].x * g, z[r].y * g)) / mag);
        }
    }
    // the rounding the kernels finish with (fft512.h round_coef): x * gain + 1.5 * 2^52 carries round(x * gain) mod 2^32 in the low
    // word of the sum, fused or not, for every |x * gain| the one-limb sums can reach (2^49.6, where FP64 is spaced 1/8 .. 1/4) as
    // long as the value is within the guard's limit of an integer -- checked against long double
    long bad = 0;
    const double magic = 6755399441055744.0;
    for (int trial = 0; trial < 2000000; trial++) {
        const int r = trial & 7;
        const double g = untwist_gain(r);
        const long long want = ((long long)rand() << 20 ^ rand()) % (1LL << 50) * ((rand() & 1) ? 1 : -1);
        const double frac = ((rand() % 2001) - 1000) * 1e-3 * 0.0625;       // distance to the integer: within the guard's limit, 1/16
        const double v = (double)(((long double)want + frac) / (long double)g);  // the pass's output before the gain
        const long long exact = llrintl((long double)v * (long double)g);
        if (exact != want) continue;  // the division's own rounding moved the value across +-0.5: not a case
        double t = fma(v, g, magic);
        unsigned lo;
        unsigned long long bits;
        __builtin_memcpy(&bits, &t, 8);
        lo = (unsigned)bits;
        if (lo != (unsigned)(unsigned long long)want) bad++;
        const double z = v * g;
        t = z + magic;
        __builtin_memcpy(&bits, &t, 8);
        if ((unsigned)bits != (unsigned)(unsigned long long)want) bad++;
    }
    printf("rounding: %ld mismatches in 2 000 000 values up to 2^50\n", bad);
    if (bad) return 1;
    printf("forward: largest absolute error %.3g (inputs < 2^7)\ninverse: largest error relative to the input magnitude %.3g\n", worst_f, worst_i);
    // forward sums are < 2^10: a few ulp of that; inverse: a few ulp relative
    return (worst_f < 1e-11 && worst_i < 1e-14) ? 0 : 1;
}
