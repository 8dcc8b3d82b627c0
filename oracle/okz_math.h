/*
 * okz_math.h -- deterministic float32 elementary functions for the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): nothing in the shipped
 * library includes this file.
 *
 * The reference evaluates these with NVIDIA libdevice / fast-math intrinsics
 * (akazed.cu:1697 exp, 1701 atan2, 1887-1888 __cosf/__sinf, 1091-1101
 * __expf/__powf).  Those approximations are proprietary and not reproducible
 * on any other platform ("parity unpinned"), so the oracle
 * defines its own: every operation below is a single IEEE-754 binary32
 * operation (+, -, *, /, fmaf, floorf, ldexpf) in a fixed order, so that a
 * HIP kernel executing the same sequence produces identical bits.
 * Compile with -ffp-contract=off.
 */
#ifndef OKZ_MATH_H
#define OKZ_MATH_H
#include <math.h>

#define OKZ_PI_F      3.14159274101257324f   /* (float)pi */
#define OKZ_HPI_F     1.57079637050628662f   /* (float)(pi/2), H_PI of cuda_utils.h:7 */
#define OKZ_PI_D      3.14159265358979323846 /* M_PI */

/* sin and cos of a (radians, |a| < ~1e4). Cody-Waite reduction by pi/2 and
 * the classic single-precision minimax polynomials on [-pi/4, pi/4]. */
static inline void okz_sincosf(float a, float* s_out, float* c_out)
{
    float q = floorf(a * 0.636619747f + 0.5f);          /* nearest multiple of pi/2 */
    float r = fmaf(q, -1.57079601287841796875f, a);    /* pi/2 hi  */
    r = fmaf(q, -3.139164786504813217e-7f, r);         /* pi/2 mid */
    r = fmaf(q, -5.390302529957764765e-15f, r);        /* pi/2 lo  */
    int n = ((int)q) & 3;
    float r2 = r * r;
    float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(sp, r2, -1.6666654611e-1f);
    float sr = fmaf(r * r2, sp, r);
    float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(cp, r2, 4.166664568298827e-2f);
    float cr = fmaf(r2 * r2, cp, fmaf(r2, -0.5f, 1.0f));
    float s, c;
    if (n == 0)      { s = sr;  c = cr;  }
    else if (n == 1) { s = cr;  c = -sr; }
    else if (n == 2) { s = -sr; c = -cr; }
    else             { s = -cr; c = sr;  }
    *s_out = s;
    *c_out = c;
}

/* atan2(y, x) in (-pi, pi]; atan2(0,0) = 0. */
static inline float okz_atan2f(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    float a = mn / mx;                                  /* in [0,1] */
    float t, base;
    if (a > 0.4142135679721832275f) {                   /* tan(pi/8) */
        t = (a - 1.0f) / (a + 1.0f);
        base = 0.785398185253143310546875f;             /* (float)(pi/4) */
    } else {
        t = a;
        base = 0.0f;
    }
    float z = t * t;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    float r = base + fmaf(p * z, t, t);
    if (ay > ax) r = OKZ_HPI_F - r;
    if (x < 0.0f) r = OKZ_PI_F - r;
    if (y < 0.0f) r = -r;
    return r;
}

/* exp(x) for float x; 0 below -87, clamped above 88. */
static inline float okz_expf(float x)
{
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float k = floorf(x * 1.44269502162933349609375f + 0.5f);
    float r = fmaf(k, -0.693359375f, x);
    r = fmaf(k, 2.12194440e-4f, r);
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float e = fmaf(p, r * r, r) + 1.0f;
    return ldexpf(e, (int)k);
}

#endif
