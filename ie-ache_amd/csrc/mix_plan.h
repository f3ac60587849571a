// The arithmetic of the rotation of roles for mid-size blind-rotation launches (evaluator.hip: plan_mix; blind_rotate_w64.hip:
// launch_mixed_phases), free of any device state so that the CPU tests can check it (ieache_debug_mix_plan).
//
// A launch of `gates` gate instances on a device of `cus` compute units (8 wave slots each) is cut into k subsets; tw of them
// at a time run on the two-waves-per-gate kernel for s2 steps while the others take s1 steps on the one-wave-per-gate kernel
// (k = 3, tw = 2 unless forced).
// One round = k phases = tw s2 + (k - tw) s1 steps for every gate.  `cycles` whole rounds, then optionally one shortened
// round (tail_s1 / tail_s2), then the ordinary slice loop for what is left (at least one step: it extracts).
#pragma once
#include <cstdint>

namespace ieache {

struct MixGeometry {
    int k = 0, tw = 0;
};

// -> false when the launch size is not one the rotation is used for.  force_k / force_tw: a forced geometry (0 = by size).
// By size: three subsets, two of them on two waves at a time, for launches of
//   * 4 .. 7 gates per CU (1 025 .. 1 792 at 256 CUs): more gates than fit on two waves each, fewer than a full round of
//     one wave each -- +2 .. 21 % over the single kernel, crossover at 7.1 per CU;
//   * 8 .. 10.5 gates per CU (2 049 .. 2 688): a full round of the one-wave kernel plus a small remainder that would cost
//     a second, nearly empty round -- +2 .. 25 %, crossover at 10.7 per CU.
// The waves of the two-wave subsets oversubscribe the CUs' eight slots there (up to 2.2 x): the hardware queues the
// workgroups, and the rotation runs at a size-independent 180-185 k gates/s, between the two kernels' own full-chip rates.
// Measured and dropped (profiles/r5_mix_sweep.txt): one of two / one of three subsets on two waves (the geometries whose
// waves FIT the slots: 3 .. 8 % slower than two of three at the same size), four subsets (four streams share the runtime's
// four hardware queues with the context's other streams).
inline bool mix_geometry_for(int64_t cus, int64_t gates, int force_k, int force_tw, MixGeometry* g) {
    if (cus <= 0 || gates <= 4 * cus) return false;  // <= 4 per CU: every gate fits on two waves
    int k = force_k, tw = force_tw;
    if (k == 0) {
        const bool below_a_round = gates <= 7 * cus, above_a_round = gates > 8 * cus && gates * 2 <= 21 * cus;
        if (!below_a_round && !above_a_round) return false;
        k = 3, tw = 2;
    } else if (k < 2 || k > 4 || tw < 1 || tw >= k) {
        return false;
    }
    g->k = k;
    g->tw = tw;
    return true;
}

struct MixSteps {
    int32_t s1 = 0, s2 = 0, cycles = 0, tail_s1 = 0, tail_s2 = 0;
    int32_t covered = 0;  // steps every gate has done after the rounds: cycles x round + the shortened round
};

// n: CMux steps of a rotation; s1: steps of a one-wave turn; ratio_x100: two-wave steps per one-wave step x 100.
// -> false when not even one whole round fits below n.
inline bool mix_steps_for(int32_t n, const MixGeometry& g, int32_t s1, int32_t ratio_x100, MixSteps* m) {
    if (s1 < 1 || ratio_x100 < 100 || g.k < 2) return false;
    const int32_t s2 = (int32_t)((int64_t)s1 * ratio_x100 / 100);
    const int32_t round = g.tw * s2 + (g.k - g.tw) * s1;
    const int32_t cycles = (n - 1) / round;  // at least one step is left for the ordinary loop
    if (cycles < 1) return false;
    m->s1 = s1;
    m->s2 = s2;
    m->cycles = cycles;
    m->tail_s1 = m->tail_s2 = 0;
    m->covered = cycles * round;
    // what the whole rounds leave is taken by one more round with both turn lengths scaled down, as long as a turn still is
    // a few steps
    const int32_t rem = n - 1 - cycles * round;
    const int32_t t1 = (int32_t)((int64_t)s1 * rem / round), t2 = (int32_t)((int64_t)s2 * rem / round);
    if (t1 >= 4 && t2 >= 4 && g.tw * t2 + (g.k - g.tw) * t1 <= rem) {
        m->tail_s1 = t1;
        m->tail_s2 = t2;
        m->covered += g.tw * t2 + (g.k - g.tw) * t1;
    }
    return true;
}

// gate instances per subset: whole workgroups of the one-wave kernel whatever its workgroup size (1 .. 4 gates)
inline int64_t mix_subset_size(int64_t gates, int k) { return ((gates + k - 1) / k + 11) / 12 * 12; }

}  // namespace ieache
