/* math_accuracy.c -- TEST INFRASTRUCTURE (not shipped, not on the product path).
 * Measures include/pworld_math.h's float32 primitives against float64 libm over EVERY float32 argument of the
 * ranges the kernels use (or every STRIDE-th one: argv[1]), and prints the table DESIGN.md section 2 quotes.
 *   gcc -O2 -fopenmp -ffp-contract=off -mfma -I include oracle/math_accuracy.c -lm -o oracle/_build/math_accuracy
 * (oracle/Makefile target `accuracy`).  Errors are relative to the float64 value; "ulp" is in units of the
 * float32 spacing at the result. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pworld_math.h"

static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

typedef struct { double rel, ulp; float at; long n; } acc;

static void report(const char *name, const char *range, acc a)
{
    printf("%-14s %-28s n = %11ld   max rel err %.3e  (%.2f ulp)  at %.9g\n", name, range, a.n, a.rel, a.ulp, (double)a.at);
}

/* fn: 0 exp, 1 log1p01, 2 softplus; arguments are the float32 values with bit patterns lo..hi (same sign) */
static acc sweep(int fn, uint32_t lo, uint32_t hi, uint32_t stride)
{
    acc best = {0, 0, 0, 0};
#pragma omp parallel
    {
        acc mine = {0, 0, 0, 0};
#pragma omp for schedule(static)
        for (int64_t b = lo; b <= (int64_t)hi; b += stride) {
            const float x = u2f((uint32_t)b);
            float got;
            double want;
            if (fn == 0) { got = pw_exp(x); want = exp((double)x); }
            else if (fn == 1) { got = pw_log1p01(x); want = log1p((double)x); }
            else { got = pw_softplus(x); want = (x > 0 ? (double)x : 0.0) + log1p(exp(-fabs((double)x))); }
            mine.n++;
            if (want == 0.0 || got == 0.0f) continue;  /* exact zeros are tested separately */
            const double rel = fabs((double)got - want) / want;
            if (rel > mine.rel) {
                mine.rel = rel;
                mine.at = x;
                int ex;
                frexp(want, &ex);
                mine.ulp = fabs((double)got - want) / ldexp(1.0, ex - 24);
            }
        }
#pragma omp critical
        {
            best.n += mine.n;
            if (mine.rel > best.rel) { best.rel = mine.rel; best.ulp = mine.ulp; best.at = mine.at; }
        }
    }
    return best;
}

int main(int argc, char **argv)
{
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1;
    printf("include/pworld_math.h (contract revision 3) against float64 libm, every %u-th float32 of each range\n", stride);
    /* pw_exp: x in (-87, 0] -- negative floats up to 87 in magnitude -- and [0, 88) */
    report("pw_exp", "x in (-87, -2^-126]", sweep(0, f2u(-1.17549435e-38f), f2u(-86.99999f), stride));
    report("pw_exp", "x in [2^-126, 88)", sweep(0, f2u(1.17549435e-38f), f2u(87.99999f), stride));
    report("pw_log1p01", "t in [2^-126, 1]", sweep(1, f2u(1.17549435e-38f), f2u(1.0f), stride));
    report("pw_softplus", "x in (-87, -2^-126]", sweep(2, f2u(-1.17549435e-38f), f2u(-86.99999f), stride));
    report("pw_softplus", "x in [2^-126, 320]", sweep(2, f2u(1.17549435e-38f), f2u(320.0f), stride));
    /* the exact values the kernels' theorems rest on */
    int ok = 1;
    ok &= f2u(pw_exp(-87.0f)) == 0 && f2u(pw_exp(-1e30f)) == 0 && f2u(pw_exp(-INFINITY)) == 0;
    ok &= f2u(pw_log1p01(0.0f)) == 0;
    ok &= f2u(pw_softplus(-87.0f)) == 0 && f2u(pw_softplus(-INFINITY)) == 0 && f2u(pw_softplus(-3e38f)) == 0;
    ok &= pw_softplus(87.0f) == 87.0f && pw_softplus(300.0f) == 300.0f && isinf(pw_softplus(INFINITY));
    ok &= isnan(pw_softplus(NAN)) && isnan(pw_exp(NAN));
    ok &= pw_exp(0.0f) == 1.0f && pw_exp(-0.0f) == 1.0f;
    printf("exact cases (zero cut at x <= -87 is +0, log1p01(0) = +0, softplus(x >= 87) = x, NaN / inf): %s\n", ok ? "ok" : "FAILED");
    return ok ? 0 : 1;
}
