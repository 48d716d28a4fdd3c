/*
 * det_trig.c -- deterministic float sin/cos for the oracle (TEST INFRASTRUCTURE ONLY).
 *
 * The reference calls MathF.Cos / MathF.Sin (CoreSLAM/CoreSLAMProcessor.cs:200-201,234-235,
 * 501-502,547-548; HectorSLAM/Matcher/ScanMatcher.cs:145-146), i.e. the platform CRT's
 * cosf/sinf -- which is not bit-reproducible across platforms (glibc vs UCRT, FMA ifuncs).
 * ORACLE_TRIG_LIBM uses this machine's libm.  ORACLE_TRIG_DET uses the routine below, which
 * the HIP path restates independently for device-side candidate generation: it evaluates
 * sin/cos in binary64 with only IEEE +,-,*,rint (no FMA, no table), then rounds ONCE to
 * binary32, so host and device agree bit-for-bit and the result equals the correctly rounded
 * cosf/sinf except when the binary64 value lies within ~1e-16 relative of a rounding
 * boundary (probability ~2^-28 per call).  tests/test_oracle_trig.py measures the agreement
 * rate with libm.
 *
 * Reduction: Cody-Waite with the classic 33+33+53-bit split of pi/2 (k*P1, k*P2 exact for
 * |k| < 2^20).  Kernels: the standard degree-13/14 minimax polynomials on [-pi/4, pi/4].
 * |a| > 65536 or non-finite falls back to libm.
 */
#include "oracle.h"
#include <math.h>

static int g_trig_mode = ORACLE_TRIG_LIBM;

void oracle_set_trig_mode(int mode) { g_trig_mode = mode; }
int  oracle_get_trig_mode(void) { return g_trig_mode; }

static const double TWO_OVER_PI = 6.36619772367581382433e-01;
static const double PIO2_1  = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
static const double PIO2_2  = 6.07710050630396597660e-11; /* next 33 bits */
static const double PIO2_2T = 2.02226624879595063154e-21; /* pi/2 - (PIO2_1 + PIO2_2) */

static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                    S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                    S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                    C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                    C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

void oracle_det_sincosf(float a, float *s, float *c)
{
    if (!(fabsf(a) <= 65536.0f)) {           /* huge / inf / NaN: libm */
        *s = sinf(a);
        *c = cosf(a);
        return;
    }
    double x = (double)a;
    double k = rint(x * TWO_OVER_PI);
    double r = ((x - k * PIO2_1) - k * PIO2_2) - k * PIO2_2T;
    double z = r * r;
    double ps = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double sn = r + (z * r) * (S1 + z * ps);
    double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double cs = 1.0 - (0.5 * z - z * pc);
    long long q = (long long)k;
    double so, co;
    switch ((int)(q & 3)) {
    case 0:  so = sn;  co = cs;  break;
    case 1:  so = cs;  co = -sn; break;
    case 2:  so = -sn; co = -cs; break;
    default: so = -cs; co = sn;  break;
    }
    *s = (float)so;
    *c = (float)co;
}

float oracle_cosf(float a)
{
    if (g_trig_mode == ORACLE_TRIG_DET) { float s, c; oracle_det_sincosf(a, &s, &c); return c; }
    return cosf(a);
}

float oracle_sinf(float a)
{
    if (g_trig_mode == ORACLE_TRIG_DET) { float s, c; oracle_det_sincosf(a, &s, &c); return s; }
    return sinf(a);
}
