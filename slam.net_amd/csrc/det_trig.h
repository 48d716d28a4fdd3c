// det_trig.h -- deterministic float sin/cos shared by the host-side C++ and the HIP kernels.
//
// The reference calls MathF.Cos/MathF.Sin (CoreSLAM/CoreSLAMProcessor.cs:234-235 and friends), i.e. the
// platform CRT, which is not bit-reproducible across platforms.  For every place where libslamhip forms
// (c, s) itself (device-side candidate generation, pose-based entry points) it uses this routine:
// binary64 Cody-Waite reduction by pi/2 (33+33+53-bit split) and the classic degree-13/14 minimax
// kernels, IEEE + - * rint only (the build uses -ffp-contract=off), one final rounding to binary32.
// The result is the correctly rounded float sin/cos except with probability ~2^-28 per call, and is
// bit-identical on host and device.  Entry points that take (px,py,c,s) leave trig to the caller.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

// |a| <= 65536 (caller's contract): the deterministic kernel alone
__host__ __device__ static inline void sh_det_sincosf_small(float a, float *s, float *c);

__host__ __device__ static inline void sh_det_sincosf(float a, float *s, float *c)
{
    if (!(fabsf(a) <= 65536.0f)) {        // huge / inf / NaN: outside the deterministic contract
        *s = sinf(a);
        *c = cosf(a);
        return;
    }
    sh_det_sincosf_small(a, s, c);
}

__host__ __device__ static inline void sh_det_sincosf_small(float a, float *s, float *c)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079632673412561417e+00, P2 = 6.07710050630396597660e-11, P2T = 2.02226624879595063154e-21;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double x = (double)a;
    double k = rint(x * TWO_OVER_PI);
    double r = ((x - k * P1) - k * P2) - k * P2T;
    double z = r * r;
    double ps = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double sn = r + (z * r) * (S1 + z * ps);
    double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double cs = 1.0 - (0.5 * z - z * pc);
    const int q = (int)k;                 // |k| <= 41722: exact
    double so, co;
    switch (q & 3) {
    case 0:  so = sn;  co = cs;  break;
    case 1:  so = cs;  co = -sn; break;
    case 2:  so = -sn; co = -cs; break;
    default: so = -cs; co = sn;  break;
    }
    *s = (float)so;
    *c = (float)co;
}

// MathEx.NormalizeAngle (BaseSLAM/MathEx.cs:116-138); C# '%' on float == fmodf (exact operation).
__host__ __device__ static inline float sh_normalize_angle(float angle)
{
    const float pi = 3.14159274f;
    float pi2 = pi * 2.0f;
    float a = fmodf(fmodf(angle, pi2) + pi2, pi2);
    if (a > pi) a -= 2.0f * pi;
    return a;
}
