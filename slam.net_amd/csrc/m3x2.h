// m3x2.h -- System.Numerics.Matrix3x2 / Vector2.Transform / Matrix4x4.Invert arithmetic, host + device.
//
// The reference builds its Hector transforms with the .NET BCL (HectorSLAM/Matcher/ScanMatcher.cs:139-142,
// Map/OccGridMap.cs:120-123, Map/GridMap.cs:46-47), which is not part of the reference tree.  This header
// restates the published dotnet/runtime v6 scalar formulas (row-vector convention) in binary32 with one
// rounding per operation; trig uses the deterministic sin/cos of det_trig.h.
#pragma once
#include "det_trig.h"

struct sh_m3x2 { float m11, m12, m21, m22, m31, m32; };

__host__ __device__ static inline sh_m3x2 sh_m3x2_rotation(float radians)
{
    const float pi = 3.14159274f;
    const float epsilon = 0.001f * pi / 180.0f;           // Matrix3x2.CreateRotation: 0.1 % of a degree
    radians = remainderf(radians, pi * 2);                // MathF.IEEERemainder
    float c, s;
    if (radians > -epsilon && radians < epsilon) { c = 1; s = 0; }
    else if (radians > pi / 2 - epsilon && radians < pi / 2 + epsilon) { c = 0; s = 1; }
    else if (radians < -pi + epsilon || radians > pi - epsilon) { c = -1; s = 0; }
    else if (radians > -pi / 2 - epsilon && radians < -pi / 2 + epsilon) { c = 0; s = -1; }
    else sh_det_sincosf(radians, &s, &c);
    sh_m3x2 r = { c, s, -s, c, 0.0f, 0.0f };
    return r;
}
__host__ __device__ static inline sh_m3x2 sh_m3x2_translation(float x, float y) { sh_m3x2 r = { 1, 0, 0, 1, x, y }; return r; }
__host__ __device__ static inline sh_m3x2 sh_m3x2_scale(float s) { sh_m3x2 r = { s, 0, 0, s, 0, 0 }; return r; }
__host__ __device__ static inline sh_m3x2 sh_m3x2_mul(sh_m3x2 a, sh_m3x2 b)
{
    sh_m3x2 m;
    m.m11 = a.m11 * b.m11 + a.m12 * b.m21;
    m.m12 = a.m11 * b.m12 + a.m12 * b.m22;
    m.m21 = a.m21 * b.m11 + a.m22 * b.m21;
    m.m22 = a.m21 * b.m12 + a.m22 * b.m22;
    m.m31 = a.m31 * b.m11 + a.m32 * b.m21 + b.m31;
    m.m32 = a.m31 * b.m12 + a.m32 * b.m22 + b.m32;
    return m;
}
__host__ __device__ static inline bool sh_m3x2_invert(sh_m3x2 m, sh_m3x2 *r)
{
    const float det = (m.m11 * m.m22) - (m.m21 * m.m12);
    if (fabsf(det) < 1.401298464e-45f) return false;      // float.Epsilon
    const float inv = 1.0f / det;
    r->m11 = m.m22 * inv;
    r->m12 = -m.m12 * inv;
    r->m21 = -m.m21 * inv;
    r->m22 = m.m11 * inv;
    r->m31 = (m.m21 * m.m32 - m.m31 * m.m22) * inv;
    r->m32 = (m.m31 * m.m12 - m.m11 * m.m32) * inv;
    return true;
}
__host__ __device__ static inline void sh_v2_transform(float x, float y, const sh_m3x2 &m, float *ox, float *oy)
{
    *ox = x * m.m11 + y * m.m21 + m.m31;
    *oy = x * m.m12 + y * m.m22 + m.m32;
}

// Matrix4x4.Invert (software path) specialised to the matcher's H: symmetric 3x3 in the upper-left block,
// zeros elsewhere, M44 = 1 (ScanMatcher.cs:198-203).  Returns the upper-left 3x3 of the inverse, row-major.
// With d = h = l = m = n = o = 0 and p = 1 the general cofactor formulas reduce to the ones below
// (the dropped terms are exact zeros, so the result is bit-identical to the general formula).
__host__ __device__ static inline bool sh_invert_h(const float H[9], float R[9])
{
    const float a = H[0], b = H[1], c = H[2];
    const float e = H[3], f = H[4], g = H[5];
    const float i = H[6], j = H[7], k = H[8];
    const float p = 1.0f, z = 0.0f;
    const float kp_lo = k * p - z * z, jp_ln = j * p - z * z, jo_kn = j * z - k * z;
    const float ip_lm = i * p - z * z, io_km = i * z - k * z, in_jm = i * z - j * z;
    const float a11 = +(f * kp_lo - g * jp_ln + z * jo_kn);
    const float a12 = -(e * kp_lo - g * ip_lm + z * io_km);
    const float a13 = +(e * jp_ln - f * ip_lm + z * in_jm);
    const float a14 = -(e * jo_kn - f * io_km + g * in_jm);
    const float det = a * a11 + b * a12 + c * a13 + z * a14;
    if (fabsf(det) < 1.401298464e-45f) return false;
    const float invDet = 1.0f / det;
    R[0] = a11 * invDet; R[3] = a12 * invDet; R[6] = a13 * invDet;
    R[1] = -(b * kp_lo - c * jp_ln + z * jo_kn) * invDet;
    R[4] = +(a * kp_lo - c * ip_lm + z * io_km) * invDet;
    R[7] = -(a * jp_ln - b * ip_lm + z * in_jm) * invDet;
    const float gp_ho = g * p - z * z, fp_hn = f * p - z * z, fo_gn = f * z - g * z;
    const float ep_hm = e * p - z * z, eo_gm = e * z - g * z, en_fm = e * z - f * z;
    R[2] = +(b * gp_ho - c * fp_hn + z * fo_gn) * invDet;
    R[5] = -(a * gp_ho - c * ep_hm + z * eo_gm) * invDet;
    R[8] = +(a * fp_hn - b * ep_hm + z * en_fm) * invDet;
    return true;
}
