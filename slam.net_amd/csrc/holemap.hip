// holemap.hip -- K2: HoleMap raster update, bit-exact and ray-order exact (gfx950 only).
//
// Replaces UpdateHoleMap (CoreSLAM/CoreSLAMProcessor.cs:496-534), DrawLaserRayOnHoleMap (:359-443) and
// ClipRay (:320-345).  The reference draws ray after ray with a read-modify-write blend
//     pix = (ushort)(((256 - alpha) * pix + alpha * pixval) >> 8)                     (:431)
// which does not commute for different pixval, so a pixel touched by several rays must see their
// fragments in ray order.  No atomics on the map and no per-pixel scratch: every pixel has exactly one
// writer, which knows all the rays that draw the pixel and blends their values in ray order.  ONE launch per update
// (k2_pixels<BUILD>; scans of more than K2_LDS_RAYS rays: k2_prepare + k2_pixels<!BUILD>):
//   tables   every workgroup, in LDS: per ray the literal float/int arithmetic of :519-530, :361-399 (clip, major axis,
//            V-profile parameters); rays are counting-sorted into 4 direction classes (major axis and its sign) x 1024
//            buckets of signed slope (minor / major).  Step x of a ray lies at major offset x and minor offset
//            m(x) = min(x, max(0, ceil((2*dyc*x - dxc) / (2*dxc)))) (closed form of the error recurrence :394-396,
//            :433-441; tests/test_closed_forms.py), and |m(x) - slope*x| <= 1/2, so only rays of a pixel's class with
//            slope in [(b-1/2)/a, (b+1/2)/a] can draw the pixel at (major a, minor b): one contiguous range of the table.
//   zone     (Chebyshev distance < 48 from the robot, where tens to a thousand rays cross a pixel) pixel-centric,
//            one / two / four pixels per wavefront: lanes test the candidate rays, hits are rank-sorted by ray index and
//            blended in that order; the few pixels with more than 64 candidates scan all rays in index order
//   beyond   one lane per (ray, step): work is proportional to what is drawn, not to the scan's bounding square.  The lane
//            looks up who else draws its pixel; the lowest ray index among the hits owns the pixel (see k2_pixels)
//   last     pixels with more hits than a lane orders are queued and drawn by the last workgroup to finish
// All integer arithmetic wraps like C# unchecked int; float->int follows cvttss2si (sh_f2i).
// Deviations from the reference (all in exception / platform-dependent territory; the CPU checker used by the tests does the same):
//   D1 non-representable pixel coordinates (NaN/inf, e.g. zero-range point) skip the ray;
//   D2 Math.Abs(int.MinValue) / int.MinValue / -1 skip the ray;
//   D4 a clipped endpoint outside the map (reachable only through int32 overflow in :329/:340) skips the ray.
#include "cs_internal.h"
#include "det_trig.h"
#include "raster.h"
#include "obstacle_dev.h"
#include <vector>
#include <atomic>
#include <algorithm>
#include <stdlib.h>

#define TS_NO_OBSTACLE 65500
#define TS_OBSTACLE 0
#define K2_NBUCK RS_NBUCK
// The zone: the Chebyshev radius round the robot inside which pixels are drawn pixel-centrically (beyond it: one lane per ray and
// step).  Where neighbouring rays lie less than a pixel apart a pixel-centric lane is several times cheaper than the ray-centric
// ones -- one lookup per pixel instead of one per crossing ray, no ownership to settle -- and a ring of radius a has 8a pixels, so
// rays overlap out to about a = rays / 8: the zone's radius follows the ray count (48 .. 192; 1080 rays: 135).  Measured at 1080
// rays on one box: radius 48 19.6 us, 96 16.9, 128 16.7, 160 16.5 (stand-alone updates, rocprofv3).
#define K2_ZONE_MIN 48
#define K2_ZONE_MAX 192
static inline __host__ __device__ int k2_zone_radius(int n_rays)
{
    const int z = n_rays / 8;
    return z < K2_ZONE_MIN ? K2_ZONE_MIN : z > K2_ZONE_MAX ? K2_ZONE_MAX : z;
}
#define K2_MAXHIT 4                    // hits a lane-per-pixel thread orders in registers
#define K2_MIXQ 192                    // zone pixels with hits inside a V that a workgroup queues in LDS for its ordered one-pixel path

// ray as the hit test takes it: clipped major length, signed clipped minor length (smin * dyc), the step beyond
// which pixval leaves TS_NO_OBSTACLE (:406), ray index (= blend order)
struct k2_cand { int dxc, sdyc, lim2, ray; };
// A ray's record, by ray index, in three parts (structure of arrays: each part is read with one aligned LDS access):
//   A  what a hit test needs: dxc, sdyc, lim2 = dx - 2*derrorv (:406), flags = valid | major_x << 1 | (smaj + 1) << 2
//   B  d = derrorv (:379/:386): with lim2 everything the V-profile's closed form needs (lim1 = lim2 + d :408, incv = -65500 / d :398)
//   C  1 / (2 * dxc) in binary64, correctly rounded: the minor offset of a step without a division (k2_minor_step)
// The sorted table (`order`: rays by direction class and slope bucket) holds ray indices only.
struct k2_vprof { int derrorv, incv, lim2, lim1; };   // (argument of the literal-capable closed form k2_pixval_closed)
struct k2_rayA { int dxc, sdyc, lim2, flags; };
struct k2_rayB { int d; };
#define K2_F_VALID 1
#define K2_F_MAJX 2

struct cs_ray {
    int valid;
    int ptr0;                 // y1*Size + x1                         (:401)
    int x1, y1;
    int dx;                   // unclipped major length after swap     (:368,:383)
    int dxc, dyc;             // clipped major / minor length          (:370-371,:384)
    int incmaj, incmin;       // ptr increments after swap             (:372-373,:385)
    int major_x;              // 1: major axis is x
    int smaj, smin;           // coordinate signs along major / minor
    int derrorv;              // :379/:386
    int lim2, lim1;           // dx - 2*derrorv, dx - derrorv          (:406,:408)
};

__device__ static inline bool clip_ray(int size, int &xyc, int &yxc, int xy, int yx)
{
    if (xyc < 0) {                                                         // :322
        if (xyc == xy) return false;                                       // :324
        int num = sh_wmul(sh_wsub(yxc, yx), sh_wsub(0, xyc));              // :329
        int den = sh_wsub(xyc, xy);
        if (den == -1 && num == INT32_MIN) return false;                   // D2
        yxc = sh_wadd(yxc, num / den);
        xyc = 0;
    }
    if (xyc >= size) {                                                     // :333
        if (xyc == xy) return false;                                       // :335
        int num = sh_wmul(sh_wsub(yxc, yx), sh_wsub(sh_wsub(size, 1), xyc)); // :340
        int den = sh_wsub(xyc, xy);
        if (den == -1 && num == INT32_MIN) return false;                   // D2
        yxc = sh_wadd(yxc, num / den);
        xyc = size - 1;
    }
    return true;
}

// (px,py,c,s) for the update pose: either given, or formed from a device-resident pose (fused path)
__device__ static inline float4 k2_pxcs(const float *d_pose, float4 h_pxcs, float scale)
{
    if (!d_pose) return h_pxcs;
    float s, c;
    sh_det_sincosf(d_pose[2], &s, &c);
    float4 q;
    q.x = d_pose[0] * scale + 0.5f;                                        // :499
    q.y = d_pose[1] * scale + 0.5f;                                        // :500
    q.z = c * scale;                                                       // :501
    q.w = s * scale;                                                       // :502
    return q;
}

__device__ static inline cs_ray k2_make_ray(const float2 p, int size, const float4 q, float scale, float hole_width)
{
    cs_ray r;
    memset(&r, 0, sizeof(r));
    const float px = q.x, py = q.y, c = q.z, s = q.w;
    const int x1 = sh_f2i(px), y1 = sh_f2i(py);                            // :505-506
    bool ok = !(x1 < 0 || x1 >= size || y1 < 0 || y1 >= size);             // :509-512 robot out of map
    float x2p = c * p.x - s * p.y;                                         // :519
    float y2p = s * p.x + c * p.y;                                         // :520
    const int xp = sh_f2i(px + x2p);                                       // :521
    const int yp = sh_f2i(py + y2p);                                       // :522
    // MathF.Sqrt is the IEEE square root.  (Not __fsqrt_rn: on this toolchain it lowers to the bare v_sqrt_f32, 1 ulp off
    // for some inputs -- found by tests/fuzz_parity.py as a ray end one pixel out.  __builtin_sqrtf without fast-math IS the
    // correctly rounded root -- v_sqrt_f32 plus a residual test of its neighbours: tools/ubench_sqrt.hip compares it with the
    // binary64 root rounded once more for every one of the 2 139 095 040 non-negative finite inputs -- and a third of the
    // binary64 detour's dependent chain, which is what a ray's making waits for since round 5.)
    const float dist = __builtin_sqrtf(x2p * x2p + y2p * y2p);             // :524
    // (:525 HoleWidth * Scale / 2 / dist: the halving is exact, so the uniform first division is a multiplication by 0.5f)
    const float add = __fdiv_rn((hole_width * scale) * 0.5f, dist);         // :525
    x2p *= (1.0f + add);                                                   // :527
    y2p *= (1.0f + add);                                                   // :528
    const int x2 = sh_f2i(px + x2p);                                       // :529
    const int y2 = sh_f2i(py + y2p);                                       // :530
    if (xp == INT32_MIN || yp == INT32_MIN || x2 == INT32_MIN || y2 == INT32_MIN) ok = false;   // D1

    int x2c = x2, y2c = y2;                                                // :361-362
    if (ok) ok = clip_ray(size, x2c, y2c, x1, y1);                         // :365
    if (ok) ok = clip_ray(size, y2c, x2c, y1, x1);                         // :366
    if (ok && (x2c < 0 || x2c >= size || y2c < 0 || y2c >= size)) ok = false;   // D4
    if (ok) {
        const int ddx = sh_wsub(x2, x1), ddy = sh_wsub(y2, y1);
        const int ddxc = x2c - x1, ddyc = y2c - y1;
        if (ddx == INT32_MIN || ddy == INT32_MIN) ok = false;              // D2
        int dx = sh_abs(ddx), dy = sh_abs(ddy);                            // :368-369
        int dxc = sh_abs(ddxc), dyc = sh_abs(ddyc);                        // :370-371
        int incmaj = sh_sign(ddx);                                         // :372
        int incmin = sh_wmul(sh_sign(ddy), size);                          // :373
        int smaj = sh_sign(ddx), smin = sh_sign(ddy), major_x = 1;
        int t;
        if (dx > dy) {                                                     // :377
            t = sh_wsub(xp, x2);                                           // :379
        } else {
            dx = dy;                                                       // :383
            int u = dxc; dxc = dyc; dyc = u;                               // :384
            u = incmaj; incmaj = incmin; incmin = u;                       // :385
            u = smaj; smaj = smin; smin = u; major_x = 0;
            t = sh_wsub(yp, y2);                                           // :386
        }
        if (t == INT32_MIN) ok = false;                                    // D2
        const int derrorv = sh_abs(t);
        if (derrorv == 0) ok = false;                                      // :389-392
        if (ok) {
            r.valid = 1;
            r.ptr0 = y1 * size + x1;                                       // :401
            r.x1 = x1; r.y1 = y1;
            r.dx = dx; r.dxc = dxc; r.dyc = dyc;
            r.incmaj = incmaj; r.incmin = incmin;
            r.major_x = major_x; r.smaj = smaj; r.smin = smin;
            r.derrorv = derrorv;
            // (incv :398 and incerrorv :399 -- an integer division -- are formed where a value inside the V is needed: k2_pixval_fast)
            r.lim2 = sh_wsub(dx, sh_wmul(2, derrorv));                     // :406
            r.lim1 = sh_wsub(dx, derrorv);                                 // :408
        }
    }
    return r;
}

// pixval at step x is the recurrence of :406-428 (the CPU checker draws it literally); here it is evaluated
// in closed form (tests/test_closed_forms.py checks it against the literal recurrence).  TS_OBSTACLE < TS_NO_OBSTACLE
// makes incerrorv <= 0, so the
// descending half (x <= lim1) never carries, and on the ascending half the carry fires on the first J steps only:
// before-correction error of step i while every step carries = u0 + i*g + d*(i-1), negative iff i*(g+d) < d - u0.
static_assert(TS_OBSTACLE < TS_NO_OBSTACLE, "k2_pixval_closed assumes a falling V-profile");
// absurd hole widths (half-width beyond 16M pixels): the closed form's intermediates could leave int32, where the
// reference's unchecked arithmetic wraps -- the recurrence (:406-428) is walked literally instead (out of line: never taken in earnest)
__device__ static __noinline__ int k2_pixval_literal(int d, int incv, int lim2, int lim1, int x)
{
    const int incerrorv = sh_wsub(TS_OBSTACLE - TS_NO_OBSTACLE, sh_wmul(d, incv));   // :399
    int pixval = TS_NO_OBSTACLE, errorv = d / 2;                       // :402,:397
    for (int xi = lim2 < 0 ? 0 : lim2 + 1; xi <= x; xi++) {
        if (xi <= lim1) {                                              // :408
            pixval = sh_wadd(pixval, incv);
            errorv = sh_wadd(errorv, incerrorv);
            if (errorv > d) { pixval = sh_wadd(pixval, -1); errorv = sh_wsub(errorv, d); }
        } else {
            pixval = sh_wsub(pixval, incv);
            errorv = sh_wsub(errorv, incerrorv);
            if (errorv < 0) { pixval = sh_wsub(pixval, -1); errorv = sh_wadd(errorv, d); }
        }
    }
    return pixval;
}
__device__ static __forceinline__ int k2_pixval_closed(const k2_vprof p, int x)
{
    if (x <= p.lim2) return TS_NO_OBSTACLE;
    const int d = p.derrorv;
    if (d > (1 << 24)) return k2_pixval_literal(d, p.incv, p.lim2, p.lim1, x);
    const int incerrorv = sh_wsub(TS_OBSTACLE - TS_NO_OBSTACLE, sh_wmul(d, p.incv));   // :399, in (-d, 0]
    const int xs = p.lim2 < 0 ? 0 : p.lim2 + 1;
    const int xm = x < p.lim1 ? x : p.lim1;
    const int n1 = xm - xs + 1 > 0 ? xm - xs + 1 : 0;                          // steps of the descending half
    const int j = (x - xs + 1) - n1;                                           // steps of the ascending half
    const int u0 = d / 2 + n1 * incerrorv, g = -incerrorv;
    int J = 0;
    if (j > 0 && d - u0 > 0) J = (d - u0 + (g + d) - 1) / (g + d) - 1;
    const int f = j < J ? j : J;
    return TS_NO_OBSTACLE + (n1 - j) * p.incv + f;                             // sincv = -1 (:374)
}
// A pixel's value inside a ray's V (x > lim2) from the ray's record: lim1 = lim2 + d (:406, :408), incv = -65500 / d (:398) -- one
// integer division where a value is needed (a twentieth of the step lanes, the marked zone pixels) instead of two in every ray's
// making, which is the table phase's critical chain.
__device__ static __forceinline__ int k2_pixval_fast(int lim2, int flags, const k2_rayB B, int x)      // x > lim2
{
    k2_vprof p; p.derrorv = B.d; p.incv = (TS_OBSTACLE - TS_NO_OBSTACLE) / B.d; p.lim2 = lim2; p.lim1 = sh_wadd(lim2, B.d);
    return k2_pixval_closed(p, x);
}
// value of step x of the ray (A, B)
__device__ static __forceinline__ int k2_value(const k2_rayA A, const k2_rayB B, int x)
{
    return x <= A.lim2 ? TS_NO_OBSTACLE : k2_pixval_fast(A.lim2, A.flags, B, x);
}
// a ray's record from the literal arithmetic of k2_make_ray
__device__ static __forceinline__ void k2_ray_record(const cs_ray &r, k2_rayA &A, k2_rayB &B, double &C)
{
    A.dxc = r.dxc; A.sdyc = r.smin * r.dyc; A.lim2 = r.lim2;
    A.flags = (r.valid ? K2_F_VALID : 0) | (r.major_x ? K2_F_MAJX : 0) | ((r.smaj + 1) << 2);
    B.d = r.derrorv;
    C = 0.0;
    if (r.valid && r.dxc > 0) {
        // 1 / (2 dxc) by the hardware reciprocal and two Newton steps: relative error below 2^-50 (k2_minor_step needs 2^-33), a dozen
        // dependent instructions instead of the IEEE division's three dozen
        const double D = 2.0 * (double)r.dxc;
        double y = __builtin_amdgcn_rcp(D);
        y = __builtin_fma(__builtin_fma(-D, y, 1.0), y, y);
        y = __builtin_fma(__builtin_fma(-D, y, 1.0), y, y);
        C = y;
    }
}
// Minor offset of step x >= 1 of a ray: m(x) = min(x, max(0, ceil((2*dyc*x - dxc) / (2*dxc)))), the closed form of the error
// recurrence (:394-396, :433-441).  Maps up to 16384 pixels a side (T = int): N = 2*dyc*x - dxc is exact in binary64, rc = 1/(2*dxc)
// correctly rounded, so N*rc is within 2^-38 of N/D (N/D < 2^14); quotients are integers or at least 1/D >= 2^-15 apart from one:
// ceil(N*rc - 2^-18) is the exact ceiling.  Five full-rate binary64 instructions instead of a float estimate with an integer
// remainder fix-up (two dozen).  Larger maps (T = long long) divide.
template <typename T>
__device__ static __forceinline__ int k2_minor_step(int x, int dxc, int dyc, double rc)
{
    if (sizeof(T) == 4) {
        const double N = __builtin_fma((double)x, (double)(2 * dyc), -(double)dxc);
        double q = __builtin_ceil(__builtin_fma(N, rc, -0x1p-18));
        q = __builtin_fmax(q, 0.0);
        q = __builtin_fmin(q, (double)x);
        return (int)q;
    }
    const T N = (T)2 * dyc * x - dxc, D = (T)2 * dxc;
    if (N <= 0) return 0;
    const T q = (N + D - 1) / D;
    return q < (T)x ? (int)q : x;
}

__device__ static __forceinline__ uint16_t k2_blend(uint16_t pix, int pixval, int alpha)
{
    return (uint16_t)(sh_wadd(sh_wmul(256 - alpha, (int)pix), sh_wmul(alpha, pixval)) >> 8);   // :431
}

// does step x = a (major offset a >= 1) of the ray draw the pixel at signed minor offset b?  (closed form, no division;
// T = int for maps up to 16384 pixels a side -- 2*dyc*a < 2^29 -- else long long: 64-bit multiplies are several
// quarter-rate instructions each)
template <typename T>
__device__ static __forceinline__ bool k2_hit(const k2_cand c, int a, int b)
{
    if (a > c.dxc) return false;
    const int B = b < 0 ? -b : b, dyc = c.sdyc < 0 ? -c.sdyc : c.sdyc;
    // the walk takes at most one minor step per major step (m(x) <= x).  With dyc <= dxc the tests below imply it; a ray
    // whose ClipRay arithmetic wrapped (:329,:340 -- an end point a million pixels out) can come back with dyc > dxc, and
    // the zone's all-rays scan, which asks every ray about every pixel, needs the cap spelled out (found by the soak)
    if (B > a) return false;
    if (B > 0 && (c.sdyc == 0 || (b > 0) != (c.sdyc > 0))) return false;
    const T N = (T)2 * dyc * a - c.dxc, D = (T)2 * c.dxc;
    if (B == 0) return N <= 0;
    if (B == a) return N > (T)(a - 1) * D;
    return N > (T)(B - 1) * D && N <= (T)B * D;
}

// counters: [0] R = longest clipped major length, [1] unused, [2] blended pixels (every step x = 0..dxc of
// a valid ray blends exactly one pixel, :404,:431), [3] x1, [4] y1, [5] robot inside the map
__device__ static __forceinline__ int k2_ray_bucket(const k2_rayA e)
{
    const int smaj = ((e.flags >> 2) & 3) - 1;
    const float tt = e.dxc > 0 ? (float)e.sdyc / (float)e.dxc : 0.0f;
    return ((e.flags & K2_F_MAJX) ? (smaj >= 0 ? 0 : 1) : (smaj >= 0 ? 2 : 3)) * K2_NBUCK + rs_bucket(tt);
}
__global__ void __launch_bounds__(1024)
k2_prepare(const float2 *__restrict__ pts, int n, int size, float scale, const float *d_pose, float4 h_pxcs, float hole_width,
           k2_rayA *__restrict__ recA, k2_rayB *__restrict__ recB, double *__restrict__ recC, int *__restrict__ order,
           int *__restrict__ start,
           int *__restrict__ counters, int *__restrict__ total_out, int *__restrict__ dirty)
{
    __shared__ int hist[4 * K2_NBUCK];
    __shared__ int wsum[16];
    __shared__ int s_R, s_total;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    for (int i = t; i < 4 * K2_NBUCK; i += 1024) hist[i] = 0;
    if (t == 0) { s_R = 0; s_total = 0; }
    __syncthreads();
    const float4 q = k2_pxcs(d_pose, h_pxcs, scale);
    int my_R = 0, my_total = 0;
    for (int i = t; i < n; i += 1024) {
        const cs_ray r = k2_make_ray(pts[i], size, q, scale, hole_width);
        k2_rayA A; k2_rayB B; double C;
        k2_ray_record(r, A, B, C);
        recA[i] = A; recB[i] = B; recC[i] = C;
        if (r.valid) {
            atomicAdd(&hist[k2_ray_bucket(A)], 1);
            my_R = max(my_R, r.dxc);
            my_total += r.dxc + 1;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {                       // one LDS atomic per wave, not per ray (same address)
        my_R = max(my_R, __shfl_down(my_R, off, 64));
        my_total += __shfl_down(my_total, off, 64);
    }
    if (lane == 0) { atomicMax(&s_R, my_R); atomicAdd(&s_total, my_total); }
    __syncthreads();
    {   // exclusive prefix over the 4096 bins: 4 consecutive bins per thread
        int v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = hist[4 * t + k]; sum += v[k]; }
        int incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int base = incl - sum;
        for (int w = 0; w < wid; w++) base += wsum[w];
#pragma unroll
        for (int k = 0; k < 4; k++) { start[4 * t + k] = base; hist[4 * t + k] = base; base += v[k]; }
        if (t == 1023) start[4 * K2_NBUCK] = base;
    }
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        const k2_rayA e = recA[i];                                 // (its own store: no other thread wrote recA[i])
        if (e.flags & K2_F_VALID) order[atomicAdd(&hist[k2_ray_bucket(e)], 1)] = i;
    }
    if (t == 0) {
        counters[0] = s_R; counters[1] = 0; counters[2] = s_total;
        if (total_out) *total_out = s_total;
        const int x1 = sh_f2i(q.x), y1 = sh_f2i(q.y);
        counters[3] = x1; counters[4] = y1;
        // the pixels this update can change lie in the scan's bounding square (the pixel kernel's very bounds): the partial
        // host mirror (slamhip_cs_holemap_mirror) copies the union of these squares since its last call
        if (dirty && s_total > 0 && x1 >= 0 && x1 < size && y1 >= 0 && y1 < size) {
            dirty[0] = min(dirty[0], max(x1 - s_R, 0)); dirty[1] = min(dirty[1], max(y1 - s_R, 0));
            dirty[2] = max(dirty[2], min(x1 + s_R, size - 1)); dirty[3] = max(dirty[3], min(y1 + s_R, size - 1));
        }
    }
}

// One wavefront draws one pixel: lanes test the candidate rays, hits are rank-sorted by ray index and blended in that
// order; the robot's pixel (step 0 of every ray) and its closest neighbours (more than 64 candidates) scan all rays in
// index order.  `sval` is 64 ints of LDS private to the wavefront.
// (recA / recB / order: LDS when the kernel made the scan's tables itself, else global)
template <typename T, typename OT, typename ST>
__device__ static __noinline__ void k2_wave_pixel(int X, int Y, int x1, int y1, int size, const k2_rayA *recA, const k2_rayB *recB,
                                            int n_rays, const OT *order, const ST *start,
                                            uint16_t *__restrict__ map, int alpha, int *sval, bool scan_all = false)
{
    const int lane = threadIdx.x & 63;
    const int ptr = Y * size + X;
    const int dx = X - x1, dy = Y - y1;
    int cls[2], a[2], b[2], lo[2] = { 0, 0 }, hi[2] = { 0, 0 };
    int ncls = 0, nc = 0;
    if (!scan_all) {                                                   // (scan_all: a core workgroup -- every ray by index, no sorted table)
        ncls = rs_classes(dx, dy, cls, a, b);
#pragma unroll
        for (int k = 0; k < 2; k++) if (k < ncls) { rs_range(start, cls[k], a[k], b[k], 0.0f, lo[k], hi[k]); nc += hi[k] - lo[k]; }
    }
    uint16_t pix = map[ptr];
    bool stable = false;
    int last_v = 0;
    // Blends of ONE value commute, and near the robot nearly every fragment carries TS_NO_OBSTACLE (step x of a ray lies below its
    // V: x <= lim2): the hits are COUNTED first, and only a pixel with a hit inside some ray's V takes the ordered path.
    if (ncls == 0 || nc > 64) {
        for (int pass = 0; pass < 2; pass++) {                      // pass 0 counts; pass 1 (a hit inside a V) blends in ray order
            const bool ordered = pass == 1;
            int nh = 0;
            unsigned long long anyv = 0ull;
            k2_rayA e_next = recA[lane < n_rays ? lane : 0];           // (the table is read one iteration ahead)
            for (int base = 0; base < n_rays; base += 64) {
                const int i = base + lane;
                bool hit = false, inv = false;
                int x = -1;
                const k2_rayA e = e_next;
                e_next = recA[i + 64 < n_rays ? i + 64 : 0];
                if (i < n_rays) {
                    if (e.flags & K2_F_VALID) {
                        const int smaj = ((e.flags >> 2) & 3) - 1;
                        const int aa = (e.flags & K2_F_MAJX) ? dx : dy, bb = (e.flags & K2_F_MAJX) ? dy : dx;
                        if (smaj != 0) x = aa * smaj; else if (aa == 0) x = 0;
                        k2_cand c; c.dxc = e.dxc; c.sdyc = e.sdyc; c.lim2 = e.lim2; c.ray = i;
                        if (x == 0 ? bb == 0 : (x > 0 && k2_hit<T>(c, x, bb))) { hit = true; inv = x > e.lim2; }
                    }
                }
                unsigned long long mask = __ballot(hit);
                if (!ordered) { nh += __popcll(mask); anyv |= __ballot(inv); continue; }
                if (mask == 0ull) continue;
                int v = TS_NO_OBSTACLE;
                if (inv) v = k2_pixval_fast(e.lim2, e.flags, recB[i], x);
                while (mask) {
                    // the leading run of hits with one value: the blend of a run converges -- once it no longer changes the
                    // pixel the rest of the run cannot either
                    const int src = __ffsll((long long)mask) - 1;
                    const int vv = __builtin_amdgcn_readlane(v, src);
                    const unsigned long long same = __ballot(hit && v == vv) & mask, diff = mask & ~same;
                    const unsigned long long run = diff ? (same & ((diff & (0ull - diff)) - 1ull)) : same;
                    for (int k = __popcll(run); k > 0 && !(stable && vv == last_v); k--) {
                        const uint16_t np = k2_blend(pix, vv, alpha);
                        stable = np == pix; pix = np; last_v = vv;
                    }
                    mask &= ~run;
                }
            }
            if (!ordered && anyv == 0ull) {
                for (int k = 0; k < nh; k++) { const uint16_t np = k2_blend(pix, TS_NO_OBSTACLE, alpha); if (np == pix) break; pix = np; }
                break;
            }
        }
    } else if (nc > 0) {
        int ci = -1, kk = 0;
        if (lane < hi[0] - lo[0]) { ci = lo[0] + lane; kk = 0; }
        else if (ncls > 1 && lane - (hi[0] - lo[0]) < hi[1] - lo[1]) { ci = lo[1] + lane - (hi[0] - lo[0]); kk = 1; }
        bool hit = false, inv = false;
        int idx = 0x7fffffff, v = TS_NO_OBSTACLE;
        if (ci >= 0) {
            const int ray = (int)order[ci];
            const k2_rayA e = recA[ray];
            k2_cand c; c.dxc = e.dxc; c.sdyc = e.sdyc; c.lim2 = e.lim2; c.ray = ray;
            const int aa = kk ? a[1] : a[0], bb = kk ? b[1] : b[0];
            if (k2_hit<T>(c, aa, bb)) { hit = true; idx = ray; inv = aa > e.lim2; if (inv) v = k2_pixval_fast(e.lim2, e.flags, recB[ray], aa); }
        }
        const unsigned long long mask = __ballot(hit);
        if (mask && __ballot(inv) == 0ull) {
            for (int k = __popcll(mask); k > 0; k--) { const uint16_t np = k2_blend(pix, TS_NO_OBSTACLE, alpha); if (np == pix) break; pix = np; }
        } else if (mask) {
            int rank = 0;
            unsigned long long m = mask;
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                const int oi = __builtin_amdgcn_readlane(idx, src);
                rank += (hit && oi < idx) ? 1 : 0;
                m &= m - 1;
            }
            if (hit) sval[rank] = v;
            __builtin_amdgcn_wave_barrier();
            const int nh = __popcll(mask);
            for (int k = 0; k < nh; k++) {
                const int vv = sval[k];
                if (!(stable && vv == last_v)) {
                    const uint16_t np = k2_blend(pix, vv, alpha);
                    stable = np == pix; pix = np; last_v = vv;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane == 0) map[ptr] = pix;
}

// ---- the pixel kernel ------------------------------------------------------------------------------------------------
// Work is proportional to what is DRAWN, not to the scan's bounding square (round 2 visited every pixel of the square: ~4 M
// lanes for ~0.64 M blended pixels at 2048^2).  Two tiers around the robot, by Chebyshev distance r (step x of a ray lies at
// r = x exactly: the walk takes at most one minor step per major step):
//   T1  r < zone           pixel-centric, wavefronts: tens to a thousand rays cross a pixel near the robot, a handful at r = 47.
//                          Pixels are numbered from the centre outwards (k2_ring_pixel); a wavefront takes one pixel (r < rB), two
//                          (rB <= r < rC: 32 lanes test the candidate rays of each) or four (16 lanes each): lanes test the candidates,
//                          hits are rank-sorted by ray index through LDS and blended in that order by the group's first lane.
//   T3  r >= zone          one lane per (ray, step): the lane computes its pixel from the closed form of the walk and asks, like a
//                          pixel-centric lane would, which rays can draw that pixel -- one contiguous range of the slope-sorted
//                          table.  Out here rays are more than a pixel apart: nearly always the range holds the lane's own ray and
//                          nothing else, and the pixel is blended at once.  Otherwise (and on the diagonals, where the two classes
//                          of a quadrant meet) the candidates are tested; the lane of the LOWEST ray index among the hits owns the
//                          pixel and blends all hits in ray order, the other hitting rays' lanes drop it.  The lookup is a function
//                          of the pixel alone, so every lane that lands on a pixel sees the same hit list and exactly one owns it.
// A pixel with more hits than a T3 lane orders in registers is drawn by its owner lane all the same (k2_lane_draw_ordered).
// The bucket table, the sorted ray table, the V-profiles (in table order) and the rays by index live in LDS: a pixel's
// lookup is a chain of dependent small reads (bucket bounds -> candidates -> V-profile -> map), which global-memory latency
// would dominate.
#ifndef K2_LDS_RAYS
#define K2_LDS_RAYS 2048               // largest scan whose tables fit the LDS: 24 KB of buckets + 44 B per ray + 20 KB of the wavefronts' selections + the kernel's static 5 KB <= 160 KB
#endif
#ifdef K2_TIMES
// developer instrumentation (build with SLAMHIP_K2_TIMES=1): 100 MHz wall-clock stamps per workgroup and phase
__device__ unsigned long long g_k2_times[512 * 8];
__device__ unsigned long long g_k2_fine[512 * 8];      // finer stamps inside the table phase: [0] points arrived + selection done [1] rays made [2] wave reductions done [3] prefix barrier passed [4] lists written
#define K2_STAMP(k) { if (threadIdx.x == 0 && blockIdx.x < 512) g_k2_times[blockIdx.x * 8 + (k)] = wall_clock64(); }
#define K2_FINE(k) { if (threadIdx.x == 0 && blockIdx.x < 512) g_k2_fine[blockIdx.x * 8 + (k)] = wall_clock64(); }
__device__ unsigned long long g_k2_sub[512 * 16 * 8];   // per wavefront: [0] T1 time [1] T1 items [2] - [3] - [4] T3 time [5] T3 items [6] longest item [7] its kind * 65536 + index
#define K2_ITEM_T0 const unsigned long long it0_ = wall_clock64();
#define K2_ITEM_T1(kind, idx) { if ((threadIdx.x & 63) == 0 && blockIdx.x < 512) { unsigned long long *p_ = g_k2_sub + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8; \
        const unsigned long long d_ = wall_clock64() - it0_; p_[2 * (kind)] += d_; p_[2 * (kind) + 1] += 1; if (d_ > p_[6]) { p_[6] = d_; p_[7] = (unsigned long long)(kind) * 65536ull + (unsigned long long)((idx) & 65535); } } }
#else
#define K2_STAMP(k) {}
#define K2_FINE(k) {}
#define K2_ITEM_T0
#define K2_ITEM_T1(kind, idx)
#endif

// pixel number i of the zone, counted from the robot's pixel outwards: ring r (Chebyshev distance r) holds the numbers
// (2r-1)^2 .. (2r+1)^2 - 1, walked along its four sides
__device__ static __forceinline__ void k2_ring_pixel(int i, int &ddx, int &ddy)
{
    ddx = 0; ddy = 0;
    if (i <= 0) return;
    int r = (int)((sqrtf((float)i) + 1.0f) * 0.5f);
    if ((2 * r - 1) * (2 * r - 1) > i) r--; else if ((2 * r + 1) * (2 * r + 1) <= i) r++;
    const int o = i - (2 * r - 1) * (2 * r - 1), side = o / (2 * r), p = o - side * 2 * r;      // 8r pixels: four sides of 2r
    ddx = side == 0 ? -r + p : side == 1 ? r : side == 2 ? r - p : -r;
    ddy = side == 0 ? -r : side == 1 ? -r + p : side == 2 ? r : r - p;
}

// One wavefront draws G = 1 << lg pixels in the ordered way, W = 64 / G lanes each: the lanes test a pixel's candidates, the hits
// are ranked by ray index and blended in that order by the group's first lane.  (ddx, ddy, exists) are a lane's group's pixel.
// A pixel with more candidates than its lanes goes down the one-pixel path, pixel after pixel: ONE call site for it per use (the
// kernel's code is executed once or twice per wavefront, from a cold instruction cache: its size is latency; nine inlined copies
// of the one-pixel path made a 47 KB kernel that ran 10 us slower than the 20 KB one).
template <typename T, typename OT, typename ST>
__device__ static __noinline__ void k2_wave_group(int ddx, int ddy, bool exists, int lg, int x1, int y1, int size, const k2_rayA *recA, const k2_rayB *recB,
                                                     int n_rays, const OT *order, const ST *start, uint16_t *__restrict__ map, int alpha, int *sval)
{
    const int W = 64 >> lg, G = 1 << lg;
    const int lane = threadIdx.x & 63, g = lane >> (6 - lg), l = lane & (W - 1);
    const int X = x1 + ddx, Y = y1 + ddy;
    const bool valid = exists && X >= 0 && X < size && Y >= 0 && Y < size;
    int cls[2], a[2], b[2], lo[2] = { 0, 0 }, hi[2] = { 0, 0 };
    int ncls = 0, nc = 0;
    if (valid && lg > 0) {
        ncls = rs_classes(ddx, ddy, cls, a, b);
#pragma unroll
        for (int k = 0; k < 2; k++) if (k < ncls) { rs_range(start, cls[k], a[k], b[k], 0.0f, lo[k], hi[k]); nc += hi[k] - lo[k]; }
    }
    if (lg == 0 || __ballot(valid && (ncls == 0 || nc > W)) != 0ull) {
        for (int gg = 0; gg < G; gg++) {
            const int Xg = __builtin_amdgcn_readlane(X, gg * W), Yg = __builtin_amdgcn_readlane(Y, gg * W);
            const int vg = __builtin_amdgcn_readlane(valid ? 1 : 0, gg * W);
            if (vg) k2_wave_pixel<T>(Xg, Yg, x1, y1, size, recA, recB, n_rays, order, start, map, alpha, sval);
        }
        return;
    }
    const int ptr = Y * size + X;
    uint16_t pix = 0;
    if (valid && l == 0) pix = map[ptr];                           // requested now, needed after the ranking
    int ci = -1, kk = 0;
    if (l < hi[0] - lo[0]) { ci = lo[0] + l; kk = 0; }
    else if (ncls > 1 && l - (hi[0] - lo[0]) < hi[1] - lo[1]) { ci = lo[1] + l - (hi[0] - lo[0]); kk = 1; }
    bool hit = false;
    int idx = 0x7fffffff, v = 0;
    if (ci >= 0) {
        const int ray = (int)order[ci];
        const k2_rayA e = recA[ray];
        k2_cand c; c.dxc = e.dxc; c.sdyc = e.sdyc; c.lim2 = e.lim2; c.ray = ray;
        const int aa = kk ? a[1] : a[0], bb = kk ? b[1] : b[0];
        if (k2_hit<T>(c, aa, bb)) { hit = true; idx = ray; v = aa <= e.lim2 ? TS_NO_OBSTACLE : k2_pixval_fast(e.lim2, e.flags, recB[ray], aa); }
    }
    const unsigned long long mask = __ballot(hit);
    if (mask == 0ull) return;
    int rank = 0;
    for (unsigned long long m = mask; m; m &= m - 1) {             // rank among the hits of the lane's own pixel
        const int src = __ffsll((long long)m) - 1;
        const int oi = __builtin_amdgcn_readlane(idx, src);
        rank += (hit && (src >> (6 - lg)) == g && oi < idx) ? 1 : 0;
    }
    if (hit) sval[g * W + rank] = v;
    __builtin_amdgcn_wave_barrier();
    const unsigned long long gm = (lg == 1 ? 0xffffffffull : 0xffffull) << (g * W);
    const int nh = __popcll(mask & gm);
    if (valid && l == 0 && nh > 0) {
        bool stable = false;
        int last_v = 0;
        for (int k0 = 0; k0 < nh; k0 += 4) {                       // (four values per LDS read: the read's latency is the loop's)
            const int4 q = *(const int4 *)&sval[g * W + k0];
            const int vv4[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
            for (int k = 0; k < 4; k++) if (k0 + k < nh && !(stable && vv4[k] == last_v)) {
                const uint16_t np = k2_blend(pix, vv4[k], alpha);
                stable = np == pix; pix = np; last_v = vv4[k];
            }
        }
        map[ptr] = pix;
    }
    __builtin_amdgcn_wave_barrier();
}

// A lane draws a pixel by itself, all hits in ray order, however many: the lowest ray index above the last one blended is looked
// for again and again (hits x candidates tests).  The way out for what the fast paths cannot hold -- a far pixel with more hits
// than a step lane orders in registers, a zone pixel that found the workgroup's queue full -- rare by construction, and any lane
// has the tables it needs (the device-wide conflict list of rounds 3 - 4 was drawn by the LAST workgroup to finish, which under arcs
// holds another octant's rays; it also put an arrival ticket, a dependent round trip, at the end of every workgroup).
// (`my_ray` >= 0: the caller is the lane of that ray on the pixel -- it draws only if no ray of a lower index hits, `owner` says so)
template <typename T, typename OT, typename ST>
__device__ static __noinline__ uint16_t k2_lane_draw_ordered(const k2_rayA *recA, const k2_rayB *recB, const OT *order, const ST *start,
                                                             int dx, int dy, uint16_t pix, int alpha, int my_ray, bool &owner)
{
    int cls[2], a[2], b[2], lo[2] = { 0, 0 }, hi[2] = { 0, 0 };
    const int ncls = rs_classes(dx, dy, cls, a, b);
#pragma unroll
    for (int k = 0; k < 2; k++) if (k < ncls) rs_range(start, cls[k], a[k], b[k], 0.0f, lo[k], hi[k]);
    owner = true;
    if (my_ray >= 0) {
#pragma unroll
        for (int k = 0; k < 2; k++) if (k < ncls) {
            for (int ci = lo[k]; ci < hi[k] && owner; ci++) {
                const int ray = (int)order[ci];
                if (ray >= my_ray) continue;
                const k2_rayA e = recA[ray];
                k2_cand c; c.dxc = e.dxc; c.sdyc = e.sdyc; c.lim2 = e.lim2; c.ray = ray;
                if (k2_hit<T>(c, a[k], b[k])) owner = false;
            }
        }
        if (!owner) return pix;
    }
    int prev = -1;
    for (;;) {
        int best = 0x7fffffff, bv = 0;
#pragma unroll
        for (int k = 0; k < 2; k++) if (k < ncls) {
            for (int ci = lo[k]; ci < hi[k]; ci++) {
                const int ray = (int)order[ci];
                if (ray <= prev || ray >= best) continue;
                const k2_rayA e = recA[ray];
                k2_cand c; c.dxc = e.dxc; c.sdyc = e.sdyc; c.lim2 = e.lim2; c.ray = ray;
                if (!k2_hit<T>(c, a[k], b[k])) continue;
                best = ray;
                bv = a[k] <= e.lim2 ? TS_NO_OBSTACLE : k2_pixval_fast(e.lim2, e.flags, recB[ray], a[k]);
            }
        }
        if (best == 0x7fffffff) break;
        pix = k2_blend(pix, bv, alpha);
        prev = best;
    }
    return pix;
}

// ---- directions as one coordinate round the robot ------------------------------------------------------------------------------
// u in [0, 8): the direction class and the signed slope sigma = minor / |major| of a pixel or ray as ONE number that is continuous
// across the class borders -- class 0 (+x): 1 + sigma, class 2 (+y): 3 - sigma, class 1 (-x): 5 - sigma, class 3 (-y): 7 + sigma -- so
// that [k, k + 1) is the k-th octant.  XCD k (the workgroups b with b % 8 == k) draws octant k: its zone pixels from rB outwards and
// the steps beyond the zone of the rays that point into it.  A pixel at major offset a is drawn only by rays whose (class, slope)
// lies within 1 / (2a) of its own u (raster.h), so a workgroup needs the rays of its octant and a margin -- and it selects them by
// the direction of the ray's FLOAT end point (two multiply-adds and a reciprocal per ray), before any of k2_make_ray's arithmetic:
//   |u(float direction) - u(class, clipped integer slope)| <= 6.1 / dxc
// (the unclipped integer deltas are the float ones + two truncations each: |s_u - t| <= (2 + 2|t|) / dx; ClipRay :320-345 moves the
// end point along the line up to a truncation in y (1 / dxc') and one in x (|s| / dxc); a ray whose class the roundings flip lies
// that close to the diagonal, where u is continuous) -- valid while nothing wraps: robot and end points within 16000 pixels, the
// hole's half width (the extension :525-530 keeps the direction for a non-negative width) below 8000, else every ray is selected.
// A ray that draws a pixel at major offset a has dxc >= a: a workgroup whose pixels start at offset a_min takes the margin
// 8 / a_min (6.1 + the pixel's own 1/2 + the bucket slack).  Who is selected only has to be a SUPERSET of who can draw.
__device__ static __forceinline__ bool k2_arc_member(const float2 p, const float4 q, float centre, float halfw)
{
    const float X = q.z * p.x - q.w * p.y, Y = q.w * p.x + q.z * p.y;      // (:519-520)
    const float ax = fabsf(X), ay = fabsf(Y);
    float u;
    if (ax > ay) { const float sg = Y * __builtin_amdgcn_rcpf(ax); u = X > 0.0f ? 1.0f + sg : 5.0f - sg; }
    else         { const float sg = X * __builtin_amdgcn_rcpf(ay); u = Y > 0.0f ? 3.0f - sg : 7.0f + sg; }
    float d = u - centre;
    d -= 8.0f * rintf(d * 0.125f);                                         // (the shorter way round)
    const bool sane = ax < 16000.0f && ay < 16000.0f;                      // (a NaN compares false: selected)
    return !sane || !(fabsf(d) > halfw);                                   // (a NaN direction: selected)
}
// octant of a ray by its exact class and slope sign (any fixed rule would do: it only says which XCD draws the ray's far steps)
__device__ static __forceinline__ int k2_ray_octant(const k2_rayA e)
{
    const int smaj = ((e.flags >> 2) & 3) - 1;
    if (e.flags & K2_F_MAJX) return smaj >= 0 ? (e.sdyc < 0 ? 0 : 1) : (e.sdyc > 0 ? 4 : 5);
    return smaj >= 0 ? (e.sdyc > 0 ? 2 : 3) : (e.sdyc < 0 ? 6 : 7);
}
// pixel number j (from 0) of octant o, counted ring by ring from ring r0: ring a holds a pixels of every octant
__device__ static __forceinline__ void k2_octant_pixel(int o, int j, int r0, int &ddx, int &ddy)
{
    const int jj = j + (r0 * (r0 - 1)) / 2;
    int a = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)jj)) * 0.5f);
    if ((a * (a - 1)) / 2 > jj) a--; else if ((a * (a + 1)) / 2 <= jj) a++;
    const int t = jj - (a * (a - 1)) / 2;
    const int bq = (o == 0 || o == 6) ? -(t + 1) : (o == 1 || o == 7) ? t : (o == 2 || o == 4) ? t + 1 : -t;    // signed minor offset
    const int cls = o >> 1;                                                // 0: +x, 1: +y, 2: -x, 3: -y  (octants in the order of u)
    ddx = cls == 0 ? a : cls == 2 ? -a : bq;
    ddy = cls == 1 ? a : cls == 3 ? -a : bq;
}

// the octant that owns the pixel at offset (dx, dy) != (0, 0) (the inverse of k2_octant_pixel: a diagonal pixel belongs to the octant
// that begins there), and the pixel's number among its octant's, ring by ring from ring 1
__device__ static __forceinline__ int k2_pixel_octant(int dx, int dy, int &num)
{
    const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy, a = adx > ady ? adx : ady;
    int o, t;
    if (adx > ady)      { if (dx > 0) { o = dy < 0 ? 0 : 1; t = dy < 0 ? -dy - 1 : dy; } else { o = dy > 0 ? 4 : 5; t = dy > 0 ? dy - 1 : -dy; } }
    else if (ady > adx) { if (dy > 0) { o = dx > 0 ? 2 : 3; t = dx > 0 ? dx - 1 : -dx; } else { o = dx < 0 ? 6 : 7; t = dx < 0 ? -dx - 1 : dx; } }
    else { o = dx > 0 ? (dy > 0 ? 2 : 0) : (dy > 0 ? 4 : 6); t = a - 1; }          // (a, a) 2 | (a, -a) 0 | (-a, a) 4 | (-a, -a) 6: the last pixel of its ring there
    num = (a * (a - 1)) / 2 + t;
    return o;
}

// One LANE draws one zone pixel (numbers pix0 .. pix0 + 63, ring by ring: the 64 pixels of an item see about the same number of
// rays).  Out from rB a pixel has a dozen or two candidates, and nearly all of a zone's fragments carry TS_NO_OBSTACLE (step x of a
// ray is below its V, x <= lim2, unless an obstacle stands within the zone's radius plus the hole's half width of the robot): blends
// of ONE value commute, so the lane only COUNTS its hits -- no ranking, no ordering -- and blends the value that many times (the
// blend converges: it stops at its fixed point).  A pixel with a hit inside some ray's V is not drawn here: it goes to the
// workgroup's queue (LDS; beyond its capacity the lane draws it itself: k2_lane_draw_ordered) and is drawn in the ordered way, four
// pixels to a wavefront, when the workgroup has finished.  3277 wavefront items of ~3 us became ~660: the zone occupies two or three wavefronts of a workgroup
// while the others draw the steps beyond it (the two phases used to run one after the other: 5.6 + 5.6 us of a 21 us launch).
template <typename T, typename OT, typename ST>
__device__ static __forceinline__ void k2_lane_pixels(int ddx, int ddy, bool exists, int x1, int y1, int size, const k2_rayA *recA,
                                                      const k2_rayB *recB, const OT *order, const ST *start, uint16_t *__restrict__ map, int alpha,
                                                      int *mixq, int *n_mixq, int cap_mixq)
{
    const int X = x1 + ddx, Y = y1 + ddy;
    if (!(exists && X >= 0 && X < size && Y >= 0 && Y < size)) return;
    const int ptr = Y * size + X;
    uint16_t pix = map[ptr];                                        // (requested now, needed after the candidates)
    int cls[2], a[2], b[2];
    const int ncls = rs_classes(ddx, ddy, cls, a, b);
    int nh = 0;
    bool mixed = false;
#pragma unroll
    for (int k = 0; k < 2; k++) if (k < ncls) {
        int lo, hi;
        rs_range(start, cls[k], a[k], b[k], 0.0f, lo, hi);
        for (int ci = lo; ci < hi; ci++) {
            const int ray = (int)order[ci];
            const k2_rayA e = recA[ray];
            k2_cand c; c.dxc = e.dxc; c.sdyc = e.sdyc; c.lim2 = e.lim2; c.ray = ray;
            if (k2_hit<T>(c, a[k], b[k])) { nh++; mixed = mixed || a[k] > e.lim2; }
        }
    }
    if (mixed) {
        // (out from radius 48 a pixel has a handful of candidates: the lane orders them itself -- hits x candidates tests; nearer the
        // robot, a dozen or two, the pixel goes to the workgroup's queue; a full queue: drawn here all the same, slowly)
        const int amax = max(ddx < 0 ? -ddx : ddx, ddy < 0 ? -ddy : ddy);
        const int slot = amax >= K2_ZONE_MIN ? cap_mixq : atomicAdd(n_mixq, 1);
        if (slot < cap_mixq) mixq[slot] = ptr;
        else { bool ow; map[ptr] = k2_lane_draw_ordered<T>(recA, recB, order, start, ddx, ddy, pix, alpha, -1, ow); }
    } else if (nh > 0) {
        for (int k = 0; k < nh; k++) {
            const uint16_t np = k2_blend(pix, TS_NO_OBSTACLE, alpha);
            if (np == pix) break;
            pix = np;
        }
        map[ptr] = pix;
    }
}

// dynamic LDS of the pixel kernel.  !BUILD: the bucket table (int).  BUILD (the kernel makes the scan's tables itself): the
// histogram / running positions of the counting sort (int: LDS atomics), the bucket table as unsigned short (a scan has at most
// K2_LDS_RAYS rays), the sorted table of ray indices (unsigned short) and the rays' records by index (16 + 16 + 8 bytes per ray)
// -- 70 KB with the kernel's static 4.3 KB at 1080 rays.
#define K2_LDS_FIXED ((4 * K2_NBUCK + 4) * 4)
#define K2_LDS_START16 ((4 * K2_NBUCK + 8) * 2)
#define K2_RPT ((K2_LDS_RAYS + 1023) / 1024)        // rays per thread of the one-launch form
#define K2_OWN_CAP ((16 * K2_RPT * 64 * 10 / 24) & ~7)   // own rays whose records fit the selection staging space (848 at two rays per thread)
static inline size_t k2_lds_bytes(bool build, int n_rays)
{
    // BUILD: + the list of the rays whose far steps the workgroup's XCD draws (2 bytes per ray) and every wavefront's selection
    // (index and point of the rays it makes: K2_RPT * 64 entries of 2 + 8 bytes), and the sorted table's slopes (4 bytes per ray)
    return build ? (size_t)4 * K2_NBUCK * 4 + K2_LDS_START16 + (size_t)4 * (size_t)((n_rays + 7) & ~7) + (size_t)32 * (size_t)((n_rays + 3) & ~3) + (size_t)16 * K2_RPT * 64 * 10
                 : (size_t)K2_LDS_FIXED;
}

// developer experiments (SLAMHIP_K2_EXP=n at build time, WRONG RESULTS): what bounds the kernel -- 1: the step lanes' stores dropped,
// 2: their loads dropped, 3: both, 4: a wavefront's loads and stores folded into one 128-byte line, 5: no zone items, 6: no step
// items, 7: tables only
#ifndef K2_EXP
#define K2_EXP 0
#endif
#if K2_EXP == 1
#define K2_EXP_LOAD(map, p) (map)[p]
#define K2_EXP_STORE(map, p, v) { if ((v) == 12345 && alpha == 77777) (map)[p] = (v); }
#elif K2_EXP == 2
#define K2_EXP_LOAD(map, p) (uint16_t)((p) & 0x7fff)
#define K2_EXP_STORE(map, p, v) (map)[p] = (v)
#elif K2_EXP == 3
#define K2_EXP_LOAD(map, p) (uint16_t)((p) & 0x7fff)
#define K2_EXP_STORE(map, p, v) { if ((v) == 12345 && alpha == 77777) (map)[p] = (v); }
#elif K2_EXP == 4
#define K2_EXP_LOAD(map, p) (map)[((p) & ~0xfff) | (threadIdx.x & 63)]
#define K2_EXP_STORE(map, p, v) (map)[((p) & ~0xfff) | (threadIdx.x & 63)] = (v)
#else
#define K2_EXP_LOAD(map, p) (map)[p]
#define K2_EXP_STORE(map, p, v) (map)[p] = (v)
#endif
// a T3 work item as a lane holds it between its fetch (the pixel's load is issued there) and its turn: step x of ray `ray`, at
// signed minor offset b
struct k2_t3 { int ptr, x, b, ray, lim2, flags, xalone; uint16_t pix; bool valid; };

// what the kernel needs of the scan when it makes the tables itself
struct k2_scan { const float2 *pts; float scale, hole_width; const float *d_pose; float4 h_pxcs; int *total_out; int *dirty; int rb_num, ncore, zone; int2 *span;
                 // the fused scan's form: the pose is not in memory yet -- the search (result-ring form: no final arriver, no chain)
                 // left only its key; every workgroup decodes the winner itself, the first one also stores the pose for later
                 // readers and delivers key + pose to the host's mailbox (k2_winner_pose)
                 const unsigned long long *win_key; const float *win_offs; int win_n_offs; float win_bx, win_by, win_bth; float *win_pose_out;
                 uint32_t *win_mail; uint32_t win_seq; };

// The winner of the search from its key, as MonteCarloSearch returns it and Update normalises it (CoreSLAMProcessor.cs:635-637, :746):
// search_pose + offs[index - 1], theta normalised.  pose[3] keeps the un-normalised theta.
__device__ static inline void k2_winner_pose(const k2_scan &sc, float pose[4], unsigned long long &key)
{
    key = __hip_atomic_load(sc.win_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t flat = (uint32_t)key;
    float bx = sc.win_bx, by = sc.win_by, th = sc.win_bth;
    if (flat > 0 && flat <= (uint32_t)sc.win_n_offs) {             // (a key nobody armed -- all ones: a rank without candidates in a group of one -- decodes to the search pose, as k_winner_from_key has it)
        bx = sc.win_bx + sc.win_offs[3 * (size_t)(flat - 1)]; by = sc.win_by + sc.win_offs[3 * (size_t)(flat - 1) + 1]; th = sc.win_bth + sc.win_offs[3 * (size_t)(flat - 1) + 2]; }
    pose[0] = bx; pose[1] = by; pose[2] = sh_normalize_angle(th); pose[3] = th;
}

// BUILD: ONE launch per HoleMap update.  Every workgroup makes the scan's ray tables itself, in LDS -- per ray the literal
// arithmetic of :519-530 / :361-399 (k2_make_ray), a counting sort into 4 direction classes x 1024 slope buckets -- instead of
// reading what a one-workgroup k2_prepare launch left in memory: 1080 rays are a microsecond of arithmetic for 1024 lanes, the
// separate launch was 7.5 us plus a launch boundary.  (The order of the rays inside a bucket differs from workgroup to workgroup
// -- LDS atomics -- so nothing that is shared out between workgroups goes by table position: T3 deals RAY INDICES.)
// !BUILD (scans of more than K2_LDS_RAYS rays): k2_prepare's tables are read from memory.
// the ObstacleMap update riding on the launch (see k2_pixels): a wavefront's 64 cells of the pending cell pass, whose loads it
// requested at its start, and its ray's walk
__device__ static __forceinline__ void k2_ride_tail(const k3_ride &ride, bool ride_cells, bool ride_ray, int ride_cell, int ride_r, int ride_nw,
                                                    uint32_t ride_h, uint8_t ride_nh, int ride_v, float2 ride_p)
{
    if (ride_cells) k3_apply_loaded(ride_cell, ride_h, ride_nh, ride_v, ride.map, ride.cell_hits, ride.cell_nohit, ride.cell_max_hits);
    for (int c = ride_cell + ride_nw * 64; c < ride.n_cells; c += ride_nw * 64)              // (ObstacleMaps of more than 512^2 cells)
        k3_apply_cell(c, ride.map, ride.n_cells, ride.cell_hits, ride.cell_nohit, ride.cell_max_hits);
    if (ride_ray) {                                                // (wave-uniform)
        const float4 qo = k3_pxcs(ride.d_pose, ride.h_pxcs, ride.scale);
        float2 pr = ride_p;
        for (int r = ride_r; r < ride.n_points; r += ride_nw) {    // (one pass: n_points <= K2_LDS_RAYS < the launch's wavefronts)
            if (r != ride_r) pr = ride.pts[r];
            const k3_walk wk = k3_walk_setup(pr, qo, ride.size);
            for (int c = 0; c < ride.chunks_per_ray && (long long)c * 64 <= wk.n; c++)
                k3_walk_iter(wk, (long long)c * 64 + (int)(threadIdx.x & 63), ride.size, ride.hits, ride.nohit);
        }
    }
}

// Row spans for the asynchronous host mirror (slamhip_cs_holemap_mirror_async): per map row the interval of columns that the
// updates since the last snapshot may have changed.  Workgroup w owns the rows [w * rpw, (w + 1) * rpw) and is their only writer;
// it walks the scan's rays (the by-index table: clipped lengths, direction flags) and, for every row of its band a ray crosses,
// the columns the ray's steps fall into there -- from the closed form of the step positions (m(k) of the table comment above):
// one column for a y-major ray, the run of an x-major one, a pixel of margin either side (the spans must COVER what was drawn,
// no more is asked of them).  Division-free for maps up to 16384 (the float reciprocal, settled exactly, as in the step lanes).
template <typename T>
__device__ static inline T k2_floor_div(T N, T D)                   // N >= 0, D > 0
{
    if (sizeof(T) == 4) {
        T q = (T)((float)N * __builtin_amdgcn_rcpf((float)D));
        T r = N - q * D;
        if (r < 0) { q--; r += D; } else if (r >= D) { q++; r -= D; }
        if (r < 0) q--; else if (r >= D) q++;
        return q;
    }
    return N / D;
}
template <typename T>
__device__ static __noinline__ void k2_row_spans(const k2_rayA *byidx, int n_rays, int x1, int y1, int size, int n_pix_wgs, int2 *__restrict__ span)
{
    __shared__ int s_lo[128], s_hi[128];
    const int t = threadIdx.x;
    const int rpw = (size + n_pix_wgs - 1) / n_pix_wgs;
    const int band0 = (int)blockIdx.x * rpw, band1 = min(band0 + rpw, size);
    for (int row0 = band0; row0 < band1; row0 += 128) {
        const int nrow = min(128, band1 - row0);
        if (t < 128) { s_lo[t] = size; s_hi[t] = -1; }
        __syncthreads();
        for (int i = t; i < n_rays; i += 1024) {
            const k2_rayA e = byidx[i];
            if (!(e.flags & 1)) continue;
            const int smaj = ((e.flags >> 2) & 3) - 1, major_x = (e.flags >> 1) & 1;
            const int dxc = e.dxc, sd = e.sdyc, dyc = sd < 0 ? -sd : sd, sgn = sd < 0 ? -1 : 1;
            const int mcap = dyc < dxc ? dyc : dxc;                // (the minor offset never exceeds the major one: m(k) <= k)
            const int ye = major_x ? y1 + sgn * mcap : y1 + smaj * dxc;
            const int ya = max(min(y1, ye), row0), yb = min(max(y1, ye), row0 + nrow - 1);
            const bool odd = dyc > dxc;                            // (a ray whose clip wrapped: covered generously)
            for (int y = ya; y <= yb; y++) {
                int xa, xb;
                if (odd) {
                    const int xe = major_x ? x1 + smaj * dxc : x1 + sgn * mcap;
                    xa = min(x1, xe); xb = max(x1, xe);
                } else if (!major_x) {                             // y-major: one pixel in the row
                    const int k = (y - y1) * smaj;
                    const T N = (T)2 * dyc * k - dxc, D = (T)2 * dxc;
                    int m = 0;
                    if (N > 0) { const T q = k2_floor_div<T>(N + D - 1, D); m = q < (T)k ? (int)q : k; }
                    xa = xb = x1 + sgn * m;
                } else {                                           // x-major: the steps k with m(k) = j
                    const int j = (y - y1) * sgn;
                    int klo = 0, khi = dxc;
                    if (dyc > 0) {
                        if (j > 0) klo = (int)k2_floor_div<T>((T)dxc * (2 * j - 1), (T)2 * dyc);
                        khi = (int)k2_floor_div<T>((T)dxc * (2 * j + 1), (T)2 * dyc) + 1;
                        if (khi > dxc) khi = dxc;
                        if (klo > dxc) klo = dxc;
                    }
                    const int p = x1 + smaj * klo, q = x1 + smaj * khi;
                    xa = min(p, q); xb = max(p, q);
                }
                xa = max(xa - 1, 0); xb = min(xb + 1, size - 1);
                atomicMin(&s_lo[y - row0], xa); atomicMax(&s_hi[y - row0], xb);
            }
        }
        __syncthreads();
        if (t < nrow && s_hi[t] >= s_lo[t]) {
            int2 g = span[row0 + t];
            g.x = min(g.x, s_lo[t]); g.y = max(g.y, s_hi[t]);
            span[row0 + t] = g;
        }
        __syncthreads();
    }
}

template <bool BUILD, typename T, bool ARCS>
__global__ void __launch_bounds__(1024)
k2_pixels(const k2_scan sc, const k2_rayA *__restrict__ recA_g, const k2_rayB *__restrict__ recB_g, const double *__restrict__ recC_g,
          const int *__restrict__ order_g, int n_rays,
          const int *__restrict__ start_g, int *__restrict__ counters, int size, uint16_t *__restrict__ map, int alpha,
          int n_pix_wgs, const k3_ride ride)
{
    extern __shared__ __attribute__((aligned(16))) char k2_smem[];
    typedef typename std::conditional<BUILD, unsigned short, int>::type start_t;
    typedef typename std::conditional<BUILD, unsigned short, int>::type order_t;
    const int n4 = (n_rays + 3) & ~3, n8 = (n_rays + 7) & ~7;
    int *pos_s = (int *)k2_smem;                                   // BUILD: histogram, then the buckets' running positions
    start_t *start = (start_t *)(k2_smem + (BUILD ? 4 * K2_NBUCK * 4 : 0));
    unsigned short *order_s = (unsigned short *)(k2_smem + 4 * K2_NBUCK * 4 + K2_LDS_START16);
    unsigned short *own_s = order_s + (BUILD ? n8 : 0);             // BUILD: the rays whose steps beyond the zone this workgroup's XCD draws
    k2_rayA *recA_s = (k2_rayA *)(own_s + (BUILD ? n8 : 0));
    k2_rayB *recB_s = (k2_rayB *)(recA_s + (BUILD ? n4 : 0));
    double *recC_s = (double *)(recB_s + (BUILD ? n4 : 0));
    float2 *selp_s = (float2 *)(recC_s + (BUILD ? n4 : 0));        // BUILD: the wavefronts' selections -- points ...
    unsigned short *seli_s = (unsigned short *)(selp_s + (BUILD ? 16 * K2_RPT * 64 : 0));   // ... and ray indices
    // (once the rays are made the same space holds the own rays' records side by side -- part A with the ray index in the flags' upper
    // half, part C -- so that a step item reads them by its position in the list: one LDS round trip instead of two)
    float *slope_s = (float *)(seli_s + (BUILD ? 16 * K2_RPT * 64 : 0));   // BUILD: the sorted table's slopes (minor / major, as bucketed), by table position
    k2_rayA *ownA_s = (k2_rayA *)selp_s;
    double *ownC_s = (double *)(ownA_s + K2_OWN_CAP);
    __shared__ __attribute__((aligned(16))) int sval[16][64];
    __shared__ __attribute__((aligned(16))) int wsum[16], wown[16];
    __shared__ int s_nextA, s_nextB, s_R, s_total, s_nmix, s_mixq[K2_MIXQ];
    __shared__ float s_wpose[4];
    K2_STAMP(0)
    // Riding along: the ObstacleMap update (obstacle_dev.h).  Every wavefront of the launch takes 64 cells of the pending cell
    // pass and (the rays going round the workgroups) at most one ray's walk: the loads are requested at the head, behind the scan
    // point and the pose the tables wait for, and the work is done when the wavefront has drawn its last pixel and would wait for
    // the rest of its workgroup -- no memory round trip and next to no time of its own.  (As extra workgroups behind the pixel
    // ones -- the launch's LDS size lets one workgroup on a CU at a time -- the ride cost 3.2 us of the fused scan's 47.7.)
    const int ride_w = (int)blockIdx.x * 16 + (int)(threadIdx.x >> 6), ride_nw = n_pix_wgs * 16;
    const int ride_r = (int)(threadIdx.x >> 6) * n_pix_wgs + (int)blockIdx.x;
    const int ride_cell = ride_w * 64 + (int)(threadIdx.x & 63);
    const bool ride_cells = BUILD && ride.on && ride_cell < ride.n_cells, ride_ray = BUILD && ride.on && ride_r < ride.n_points;
    uint32_t ride_h = 0; uint8_t ride_nh = 0; int ride_v = 0;
    float2 ride_p = make_float2(0.f, 0.f);
    // Who draws what (BUILD with sc.ncore > 0: "arcs").  Workgroup b runs on XCD b % 8.  The first sc.ncore workgroups of every XCD
    // are CORE workgroups: tables of all rays, the zone's central pixels (one per wavefront, r < rB: a window there spans up to the
    // whole circle), no far steps.  The others are SECTOR workgroups of their XCD's octant: tables of the rays of that octant and a
    // margin (k2_arc_member), the octant's zone pixels from rB on (one per lane) and the far steps of the octant's rays.  Without
    // arcs (large scans whose tables k2_prepare made, a host mirror's row spans -- k2_row_spans walks every ray --, developer grids,
    // absurd hole widths) every workgroup holds every ray and everything is dealt round-robin, as before round 5.
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_in_xcd = (n_pix_wgs - xcd + 7) >> 3;
    constexpr bool arcs = BUILD && ARCS;                            // (a template parameter: each form's dead paths drop out of its code -- the kernel's size is latency)
    const bool is_core = !arcs || wg_in_xcd < sc.ncore;
    const bool core_count = arcs && is_core;                        // this workgroup draws its octant's central pixels by counting (below)
    int rB = (sc.rb_num * n_rays + 1079) / 1080;                    // (the radius from which a zone pixel is one lane's: by the ray COUNT, so that every workgroup agrees)
    const int zone = sc.zone > 0 ? sc.zone : k2_zone_radius(n_rays);   // (steps below `zone`: pixel-centric; from `zone` on: the step lanes)
    rB = rB < 1 ? 1 : rB > K2_ZONE_MIN ? K2_ZONE_MIN : rB;
    int R, x1, y1, n_own = 0;
    if (BUILD) {
        const int t = threadIdx.x, lane_ = t & 63, wid = t >> 6;
        constexpr int RPT = K2_RPT;                                 // rays per thread
        float2 pa[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) { const int i = t + k * 1024; pa[k] = i < n_rays ? sc.pts[i] : make_float2(0.f, 0.f); }
        float4 q;
        if (sc.win_key) {                                          // (uniform)
            float wp[4]; unsigned long long wkey;
            k2_winner_pose(sc, wp, wkey);
            float s_, c_;
            sh_det_sincosf(wp[2], &s_, &c_);
            q.x = wp[0] * sc.scale + 0.5f; q.y = wp[1] * sc.scale + 0.5f; q.z = c_ * sc.scale; q.w = s_ * sc.scale;   // :499-502, as k2_pxcs
            if (t == 0) { s_wpose[0] = wp[0]; s_wpose[1] = wp[1]; s_wpose[2] = wp[2]; s_wpose[3] = wp[3]; }          // (the ride reads it behind the barriers below)
            if (blockIdx.x == 0 && t == 0) {
                sc.win_pose_out[0] = wp[0]; sc.win_pose_out[1] = wp[1]; sc.win_pose_out[2] = wp[2]; sc.win_pose_out[3] = wp[3];
                if (sc.win_mail) {                                 // key and pose into the context's mailbox, then the completion word (common.h)
                    *(unsigned long long *)sc.win_mail = wkey;
                    float *mp = (float *)(sc.win_mail + 2); mp[0] = wp[0]; mp[1] = wp[1]; mp[2] = wp[2];
                    __hip_atomic_store(sc.win_mail + 15, sc.win_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        } else q = k2_pxcs(sc.d_pose, sc.h_pxcs, sc.scale);
        // (the ride's loads, behind the ones the tables wait for; consumed when the wavefront has drawn its last pixel)
        if (ride_cells) { ride_h = ride.cell_hits[ride_cell]; ride_nh = ride.cell_nohit[ride_cell]; ride_v = ride.map[ride_cell]; }
        if (ride_ray) ride_p = ride.pts[ride_r];
        for (int i = t; i < 4 * K2_NBUCK; i += 1024) pos_s[i] = 0;  // (the histogram, then the running positions)
        if (t == 0) { s_nextA = 0; s_nextB = 0; s_R = 0; s_total = 0; s_nmix = 0; }
        __syncthreads();
        K2_STAMP(6)
        // selection: the rays this workgroup needs, compacted INSIDE every wavefront (its threads' rays of all passes into its own
        // stretch of LDS: no barrier, no atomics; the records go by ray index, so it does not matter who makes a ray) -- a sector
        // workgroup's wavefronts hold about 40 selected rays of their 128 and make them in ONE pass (the 56 rays beyond 1024 of a
        // 1080-ray scan used to be a second pass of the first wavefront alone: 1.3 us of the table phase)
        float2 *wsel_p = selp_s + wid * (RPT * 64);
        unsigned short *wsel_i = seli_s + wid * (RPT * 64);
        int n_sel = 0;                                              // (of this wavefront)
        {
            const float centre = (float)xcd + 0.5f, halfw = 0.5f + 8.0f / (float)rB + 0.02f;
            const bool all = is_core || !(fabsf(q.x) < 16000.0f && fabsf(q.y) < 16000.0f);
#pragma unroll
            for (int k = 0; k < RPT; k++) {
                const int i = t + k * 1024;
                if (k * 1024 < n_rays) {                            // (uniform)
                    const bool in = i < n_rays, member = in && (all || k2_arc_member(pa[k], q, centre, halfw));
                    const unsigned long long mb = __ballot(member);
                    const int slot = n_sel + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u));
                    if (member) { wsel_i[slot] = (unsigned short)i; wsel_p[slot] = pa[k]; }
                    n_sel += (int)__popcll(mb);
                    if (in && !member) { k2_rayA z; z.dxc = 0; z.sdyc = 0; z.lim2 = 0; z.flags = 0; recA_s[i] = z; }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                            // (the wavefront's own LDS writes, read below by other lanes: in order)
        K2_FINE(0)
        int bkt[RPT], bray[RPT];                                    // a selected ray's bucket (class * 1024 + slope bucket; -1: not valid) and index
        int my_R = 0, my_total = 0;
#pragma unroll
        for (int k = 0; k < RPT; k++) { bkt[k] = -1; bray[k] = 0; }
#ifdef K2_TIMES
        const unsigned long long tm0_ = wall_clock64();
#endif
#pragma unroll 1
        for (int it = 0; it * 64 < n_sel; it++) {                   // (rolled: k2_make_ray is kilobytes of code, fetched once per launch)
            const int j = lane_ + it * 64;
            int bb = -1, ri = 0;
            if (j < n_sel) {
                ri = (int)wsel_i[j];
                const cs_ray r = k2_make_ray(wsel_p[j], size, q, sc.scale, sc.hole_width);
                k2_rayA ee; k2_rayB eb; double ec;
                k2_ray_record(r, ee, eb, ec);
                recA_s[ri] = ee; recB_s[ri] = eb; recC_s[ri] = ec;
                if (r.valid) {
                    if (!core_count) { bb = k2_ray_bucket(ee); atomicAdd(&pos_s[bb], 1); }     // (a counting core workgroup sorts nothing)
                    my_R = max(my_R, r.dxc);
                    my_total += r.dxc + 1;
                }
            }
#pragma unroll
            for (int k = 0; k < RPT; k++) if (k == it) { bkt[k] = bb; bray[k] = ri; }
        }
        K2_FINE(1)
#ifdef K2_TIMES
        if (lane_ == 0 && blockIdx.x < 512) { unsigned long long *p_ = g_k2_sub + ((size_t)blockIdx.x * 16 + wid) * 8; p_[2] = wall_clock64() - tm0_; p_[3] = (unsigned long long)n_sel; }
#endif
        my_R = sh_wave_max_to_lane63(my_R);                         // one LDS atomic per wave, not per ray (same address)
        my_total = sh_wave_scan_incl(my_total);
        if (lane_ == 63) { atomicMax(&s_R, my_R); atomicAdd(&s_total, my_total); }
        K2_FINE(2)
        __syncthreads();
        K2_STAMP(7)
        unsigned long long ownb[K2_RPT];
        k2_rayA ownrec[K2_RPT];
        int ownpos[K2_RPT];
#pragma unroll
        for (int k = 0; k < K2_RPT; k++) { ownb[k] = 0ull; ownpos[k] = -1; ownrec[k].dxc = 0; ownrec[k].sdyc = 0; ownrec[k].lim2 = 0; ownrec[k].flags = 0; }
        if (!core_count) {
        // the rays whose far steps this workgroup's XCD draws: with arcs the octant's (none for a core workgroup), else the XCD's
        // eighth of the scan by index -- valid rays that reach beyond the zone, in an order every workgroup of the XCD agrees on
        // (wavefront, pass, lane: the threads' rays are the same in all of them)
        int own_cnt = 0;
        {
            const int c0 = (int)(((long long)n_rays * xcd) >> 3), c1 = (int)(((long long)n_rays * (xcd + 1)) >> 3);
#pragma unroll
            for (int k = 0; k < RPT; k++) {
                const int i = t + k * 1024;
                bool own = false;
                k2_rayA e; e.dxc = 0; e.sdyc = 0; e.lim2 = 0; e.flags = 0;
                if (i < n_rays && !(arcs && is_core)) {
                    e = recA_s[i];
                    own = (e.flags & K2_F_VALID) && e.dxc >= zone && (arcs ? k2_ray_octant(e) == xcd : (i >= c0 && i < c1));
                }
                ownrec[k] = e;
                ownb[k] = __ballot(own);
                own_cnt += (int)__popcll(ownb[k]);
            }
            if (lane_ == 0) wown[wid] = own_cnt;
        }
        {   // exclusive prefix over the 4096 bins: 4 consecutive bins per thread
            int v[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = pos_s[4 * t + k]; sum += v[k]; }
            const int incl = sh_wave_scan_incl(sum);
            if (lane_ == 63) wsum[wid] = incl;
            __syncthreads();
            K2_FINE(3)
            int base = incl - sum;
#pragma unroll
            for (int w4 = 0; w4 < 4; w4++) {                          // (the sixteen wave sums in four 16-byte reads)
                const int4 ws = *(const int4 *)&wsum[4 * w4];
                base += (4 * w4 < wid ? ws.x : 0) + (4 * w4 + 1 < wid ? ws.y : 0) + (4 * w4 + 2 < wid ? ws.z : 0) + (4 * w4 + 3 < wid ? ws.w : 0);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) { start[4 * t + k] = (start_t)base; pos_s[4 * t + k] = base; base += v[k]; }     // (each thread its own four bins)
            if (t == 1023) start[4 * K2_NBUCK] = (start_t)base;
        }
        {
            int obase = 0;
#pragma unroll
            for (int w4 = 0; w4 < 4; w4++) {
                const int4 ws = *(const int4 *)&wown[4 * w4];
                n_own += ws.x + ws.y + ws.z + ws.w;
                obase += (4 * w4 < wid ? ws.x : 0) + (4 * w4 + 1 < wid ? ws.y : 0) + (4 * w4 + 2 < wid ? ws.z : 0) + (4 * w4 + 3 < wid ? ws.w : 0);
            }
#pragma unroll
            for (int k = 0; k < RPT; k++) {
                if ((ownb[k] >> lane_) & 1ull) {
                    const int i = t + k * 1024, pos = obase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ownb[k] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ownb[k], 0u));
                    own_s[pos] = (unsigned short)i; ownpos[k] = pos;
                    if (n_own <= K2_OWN_CAP) { k2_rayA w = ownrec[k]; w.flags |= i << 16; ownA_s[pos] = w; ownC_s[pos] = recC_s[i]; }
                }
                obase += (int)__popcll(ownb[k]);
            }
        }
        __syncthreads();
        K2_FINE(4)
#pragma unroll
        for (int it = 0; it < RPT; it++)
            if (bkt[it] >= 0) {
                const int pos = atomicAdd(&pos_s[bkt[it]], 1);
                const k2_rayA e = recA_s[bray[it]];                 // (this thread's own store)
                order_s[pos] = (unsigned short)bray[it];
                slope_s[pos] = e.dxc > 0 ? (float)e.sdyc / (float)e.dxc : 0.0f;
            }
        }
        R = s_R; x1 = sh_f2i(q.x); y1 = sh_f2i(q.y);
        if (blockIdx.x == 0 && t == 0) {                           // what the host reads: reach, blended pixels, the robot's pixel (workgroup 0 holds every ray)
            counters[0] = R; counters[2] = s_total; counters[3] = x1; counters[4] = y1;
            if (sc.total_out) *sc.total_out = s_total;
            // the pixels this update can change lie in the scan's bounding square: the partial host mirror
            // (slamhip_cs_holemap_mirror) copies the union of these squares since its last call
            if (sc.dirty && s_total > 0 && x1 >= 0 && x1 < size && y1 >= 0 && y1 < size) {
                sc.dirty[0] = min(sc.dirty[0], max(x1 - R, 0)); sc.dirty[1] = min(sc.dirty[1], max(y1 - R, 0));
                sc.dirty[2] = max(sc.dirty[2], min(x1 + R, size - 1)); sc.dirty[3] = max(sc.dirty[3], min(y1 + R, size - 1));
            }
        }
        __syncthreads();
        // From which step on is an own ray ALONE on its pixels?  Two rays of a class share a pixel at major offset a only if their
        // slopes differ by at most 1 / a (both minor offsets lie within 1/2 of slope * a): the ray's thread looks at the slopes of
        // the eleven buckets either side of its own, takes the smallest difference g, and from step 1 / (g - 4e-7) + 2 on the ray's
        // step lanes need no range lookup -- candidate ranges are whole buckets, and at 600 pixels nearly half the lanes used to find
        // company there that never draws their pixel, which sent almost every item down the slow part.  No neighbour in sight:
        // g >= 10 buckets.  (Diagonal pixels -- the other class of the quadrant draws there -- never take the shortcut.)
        if (n_own > 0 && n_own <= K2_OWN_CAP) {                     // (uniform)
#pragma unroll
            for (int k = 0; k < RPT; k++) if (ownpos[k] >= 0) {
                const k2_rayA e = ownrec[k];
                const int bk = k2_ray_bucket(e), cb = bk & ~(K2_NBUCK - 1);
                const int w0 = (int)start[max(bk - 11, cb)], w1 = (int)start[min(bk + 11, cb + K2_NBUCK - 1) + 1];
                const float sl = e.dxc > 0 ? (float)e.sdyc / (float)e.dxc : 0.0f;
                float g = 10.0f * (2.0f / (float)K2_NBUCK);
                int same = 0;                                       // (entries with the ray's very slope: its own, and any other -> never alone)
                for (int ci = w0; ci < w1; ci++) {
                    const float d = fabsf(slope_s[ci] - sl);
                    same += d == 0.0f ? 1 : 0;
                    g = d > 0.0f && d < g ? d : g;
                }
                const int xa = same == 1 && g > 1.0e-6f ? min((int)(__builtin_amdgcn_rcpf(g - 4.0e-7f) * 1.0001f) + 2, 2047) : 2047;
                ownA_s[ownpos[k]].flags |= xa << 5;                 // (bits 5 .. 15 were zero: "not known yet" = no shortcut)
            }
            // (no barrier: a step item fetched before its ray's thread got here reads zero there and takes the range lookup -- slower,
            // never wrong; the word is written once, by one thread)
        }
        if (x1 < 0 || x1 >= size || y1 < 0 || y1 >= size) {         // robot outside the map: nothing is drawn (:509-512)
            if (ride.on) { k3_ride rd = ride; if (sc.win_key) rd.d_pose = s_wpose; k2_ride_tail(rd, ride_cells, ride_ray, ride_cell, ride_r, ride_nw, ride_h, ride_nh, ride_v, ride_p); }   // (the ObstacleMap has its own test, at its own scale :557-560)
            return;
        }
    } else {
        R = counters[0]; x1 = counters[3]; y1 = counters[4];
        if (x1 < 0 || x1 >= size || y1 < 0 || y1 >= size) return;   // robot outside the map: nothing is drawn (:509-512)
        if (threadIdx.x == 0) { s_nextA = 0; s_nextB = 0; s_nmix = 0; }
        for (int i = threadIdx.x; i <= 4 * K2_NBUCK; i += 1024) start[i] = (start_t)start_g[i];
        __syncthreads();
    }
    K2_STAMP(1)
    const k2_rayA *recA = BUILD ? recA_s : recA_g;
    const k2_rayB *recB = BUILD ? recB_s : recB_g;
    const double *recC = BUILD ? recC_s : recC_g;
    const order_t *order = BUILD ? (const order_t *)order_s : (const order_t *)order_g;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Work items are dealt to the workgroups round-robin and inside a workgroup to whichever wavefront is free (an LDS
    // counter): an item costs what its pixels' hit lists cost, and a workgroup is only as fast as its slowest wavefront.
    // The zone: the central pixels (Chebyshev radius < rB, numbered from the robot's pixel outwards -- the closer, the more rays cross
    // a pixel: the longest items start first) are one wavefront's each; from rB on one lane's (k2_lane_pixels).
    const int Z = zone - 1 < R ? zone - 1 : R, n_pix = (2 * Z + 1) * (2 * Z + 1);
    const int pA = min((2 * rB - 1) * (2 * rB - 1), n_pix);
    const int nA = pA;                                              // central pixels: one item each
    const int nL = (n_pix - pA + 63) / 64;                          // no arcs: the rest of the zone in ring order, 64 pixels an item
    const int npo = Z >= rB ? (Z * (Z + 1) - rB * (rB - 1)) / 2 : 0, nLo = (npo + 63) / 64;    // arcs: an octant's pixels of the rings rB .. Z
    // (the dealing: a core workgroup's items are the central ones, a sector workgroup's its octant's; without arcs everything goes round)
    const int zone_first = !arcs ? (int)blockIdx.x : is_core ? wg_in_xcd * 8 + xcd : wg_in_xcd - sc.ncore;
    const int zone_step = !arcs ? n_pix_wgs : is_core ? sc.ncore * 8 : wgs_in_xcd - sc.ncore;
    const int zone_items = !arcs ? nA + nL : is_core ? 0 : nLo;          // (a core workgroup's central pixels: by counting, below)
    // T3: one lane per (ray, step) beyond the zone, ray = an entry of the sorted table, steps in blocks of 64.  Software
    // pipeline: an item's pixel is requested when the item is fetched, TWO iterations before its turn -- the map sits in HBM /
    // Infinity Cache, a microsecond away, and the phase is bound by that latency, not by its instructions (an item took 1.15 us with
    // one item in flight per wavefront whether its lanes ran 250 or 60 instructions).
    // (Measured and rejected, round 3: sectors of equal WORK instead of equal counts -- a prefix sum over the rays' blocks of 64
    // steps in the table phase, bounds where it passes k/8 of the total, a sector's blocks ending with its own longest ray.  On the
    // benchmark scan the equal-count sectors hold 574 .. 1620 non-empty blocks, but their XCDs finish within 1.5 us of each other
    // -- a block near the robot, where rays lie a pixel apart and pixels have several candidates, costs several times one far out --
    // and the equal-work form was no better balanced and paid 3 us in the table phase.)
    // Rays are dealt BY INDEX (a scan's rays come in order of their angle: neighbours in index are neighbours in direction), and to
    // the XCDs by sector -- XCD s (workgroup b runs on XCD b % 8) draws the s-th eighth of the scan: a ray's pixels share their
    // 128-byte lines with its neighbours' (at r = 600 px adjacent rays are 3.5 px apart), and a line should meet one L2.
    const int nblk = R >= zone ? (R - zone) / 64 + 1 : 0;            // steps zone .. R
    const int c0 = (int)(((long long)n_rays * xcd) >> 3);            // (!BUILD: the XCD's eighth of the scan by index; BUILD: the list own_s)
    const int n_sec = BUILD ? n_own : (int)(((long long)n_rays * (xcd + 1)) >> 3) - c0;
    const int t3_first = arcs ? wg_in_xcd - sc.ncore : wg_in_xcd, t3_step = arcs ? wgs_in_xcd - sc.ncore : wgs_in_xcd;
    const int n_t3 = (K2_EXP == 6 || K2_EXP == 7) ? 0 : nblk * n_sec;      // (K2_EXP 5 / 6 / 7: no zone items / no step items / tables only)
    const float rcp_nv = __builtin_amdgcn_rcpf((float)(n_sec > 0 ? n_sec : 1));
    // The step lanes are bound by their instruction count (2.8 M vector instructions of a launch's 5 M were theirs: 5 us of issue
    // time on the sector workgroups' SIMDs): an item whose block lies beyond its ray's end leaves at once (a third of them), a lane
    // beyond the step from which its ray is alone (xalone, table phase) neither looks up a range nor tests anything, and only the
    // blocks near the robot, where neighbouring rays are less than a pixel apart, take the slow part.
    const bool own_direct = BUILD && n_own <= K2_OWN_CAP;
#define K2_FETCH(it, more_)                                                                              \
    {                                                                                                   \
        int k_;                                                                                         \
        SH_WAVE_FETCH(k_, atomicAdd(&s_nextB, 1))                                                       \
        const int item_ = t3_first + k_ * t3_step;                                                      \
        more_ = item_ < n_t3;                                                                           \
        (it).valid = false;                                                                             \
        if (more_) {                                                                                    \
            int blk_ = (int)((float)item_ * rcp_nv);       /* item / n_sec (item < 2^24: settled exactly below) */ \
            int ri_ = item_ - blk_ * n_sec;                                                             \
            if (ri_ < 0) { blk_--; ri_ += n_sec; } else if (ri_ >= n_sec) { blk_++; ri_ -= n_sec; }     \
            k2_rayA me_; int ray_;                         /* (uniform: LDS broadcasts) */               \
            if (own_direct) { me_ = ownA_s[ri_]; ray_ = (int)((unsigned)me_.flags >> 16); }              \
            else { ray_ = BUILD ? (int)own_s[ri_] : ri_ + c0; me_ = recA[ray_]; }                        \
            const int x0_ = zone + blk_ * 64;                                                           \
            if ((me_.flags & K2_F_VALID) && x0_ <= me_.dxc) {              /* (uniform: the block holds steps of the ray) */ \
                const double rc_ = own_direct ? ownC_s[ri_] : recC[ray_];                               \
                const int x_ = x0_ + lane;                                                              \
                if (x_ <= me_.dxc) {                                                                    \
                    const int smaj_ = ((me_.flags >> 2) & 3) - 1;                                       \
                    const int dyc_ = me_.sdyc < 0 ? -me_.sdyc : me_.sdyc;                               \
                    const int m_ = k2_minor_step<T>(x_, me_.dxc, dyc_, rc_);                            \
                    const int b_ = me_.sdyc < 0 ? -m_ : m_, a_ = smaj_ < 0 ? -x_ : x_;                  \
                    const int dx_ = (me_.flags & K2_F_MAJX) ? a_ : b_, dy_ = (me_.flags & K2_F_MAJX) ? b_ : a_; \
                    /* (from which step on the ray is alone on its pixels: bits 5 .. 15 of an own record's flags, 2047: never) */ \
                    const int xa_ = own_direct ? (me_.flags >> 5) & 2047 : 2047;       /* (0: not known yet) */ \
                    (it).x = x_; (it).b = b_; (it).ray = ray_; (it).lim2 = me_.lim2; (it).flags = me_.flags & 31; (it).valid = true; \
                    (it).xalone = (xa_ == 2047 || xa_ == 0) ? 0x7fffffff : xa_;                         \
                    (it).ptr = (y1 + dy_) * size + (x1 + dx_);         /* (step pixels of a clipped ray lie inside the map) */ \
                    (it).pix = K2_EXP_LOAD(map, (it).ptr);                                              \
                }                                                                                       \
            }                                                                                           \
        }                                                                                               \
    }
    // (the first item's pixels are requested before the zone is drawn: disjoint pixels -- steps below `zone` there, from
    // `zone` on here -- and the zone's dependent chains hide the map's latency)
    k2_t3 cur, nxt;
    cur.ptr = 0; cur.x = K2_ZONE_MIN; cur.b = cur.ray = cur.lim2 = cur.flags = 0; cur.xalone = 0x7fffffff; cur.pix = 0; cur.valid = false; nxt = cur;
    bool more0 = false, more1 = false;
    if (n_t3 > 0) K2_FETCH(cur, more0)
    // A core workgroup (arcs) draws its octant's CENTRAL pixels -- the rings below rB, where a pixel's window spans up to the whole
    // circle -- by COUNTING, ray-centrically: every step x < rB of every ray that points into the octant or one next to it (a step's
    // pixel lies in the ray's own octant or an adjacent one) adds one to its pixel's counter in LDS; blends of one value commute, and
    // below a ray's V (x <= lim2) the value is TS_NO_OBSTACLE, so a pixel's counter is all it needs.  A step inside a V marks its
    // pixel instead, and marked pixels (an obstacle within rB pixels plus the hole's half width of the robot) are drawn in the
    // ordered way, one wavefront each scanning the rays by index.  529 wavefront items of 3 - 7 us (the launch's critical chain,
    // behind a full sort) became 13 000 lane steps: no sorted table in a core workgroup at all.
    if (core_count && K2_EXP != 5 && K2_EXP != 7) {
        const int t = threadIdx.x;
        int *cnt = pos_s;                                          // (zero since the launch's first barrier: a core workgroup makes no histogram)
        int *cq = pos_s + 2048;                                    // marked pixels
        const int rc_max = rB - 1 < Z ? rB - 1 : Z;                 // central rings 1 .. rc_max
        int nv = 0, mix0 = 0;
#pragma unroll
        for (int k = 0; k < K2_RPT; k++) {
            const int i = t + k * 1024;
            if (i < n_rays) {
                const k2_rayA e = recA[i];
                if (e.flags & K2_F_VALID) {
                    nv++; mix0 |= e.lim2 < 0 ? 1 : 0;
                    const int d8 = (k2_ray_octant(e) - xcd) & 7;
                    if (d8 == 0 || d8 == 1 || d8 == 7) {
                        const int smaj = ((e.flags >> 2) & 3) - 1, dyc = e.sdyc < 0 ? -e.sdyc : e.sdyc;
                        const double rc = recC[i];
                        const int xe = rc_max < e.dxc ? rc_max : e.dxc;
                        for (int x = 1; x <= xe; x++) {
                            const int m = k2_minor_step<T>(x, e.dxc, dyc, rc);
                            const int bq = e.sdyc < 0 ? -m : m, aq = smaj < 0 ? -x : x;
                            int num;
                            if (k2_pixel_octant((e.flags & K2_F_MAJX) ? aq : bq, (e.flags & K2_F_MAJX) ? bq : aq, num) == xcd) {
                                if (x <= e.lim2) atomicAdd(&cnt[num], 1); else atomicOr(&cnt[num], 1 << 30);
                            }
                        }
                    }
                }
            }
        }
        if (xcd == 0) {                                            // the robot's own pixel: step 0 of every ray
            const int lane_ = t & 63;
            const int tot = sh_wave_scan_incl(nv);
            const unsigned long long mb = __ballot(mix0 != 0);
            if (lane_ == 63) { atomicAdd(&cnt[2047], tot); if (mb) atomicOr(&cnt[2047], 1 << 30); }
        }
        __syncthreads();
        const int npc = (rc_max * (rc_max + 1)) / 2 + (xcd == 0 ? 1 : 0);
        for (int j = t; j < npc; j += 1024) {
            int ddx = 0, ddy = 0;
            const bool robot = j == (rc_max * (rc_max + 1)) / 2;
            if (!robot) k2_octant_pixel(xcd, j, 1, ddx, ddy);
            const int X = x1 + ddx, Y = y1 + ddy, c = cnt[robot ? 2047 : j];
            if (c != 0 && X >= 0 && X < size && Y >= 0 && Y < size) {
                const int ptr = Y * size + X;
                if (c >> 30) cq[atomicAdd(&s_nmix, 1)] = ptr;
                else {
                    uint16_t pix = map[ptr];
                    for (int k = 0; k < c; k++) { const uint16_t np = k2_blend(pix, TS_NO_OBSTACLE, alpha); if (np == pix) break; pix = np; }
                    map[ptr] = pix;
                }
            }
        }
    }
    for (;;) {
        int k;
        SH_WAVE_FETCH(k, atomicAdd(&s_nextA, 1))
        const int item = zone_first + k * zone_step;
        if (item >= zone_items || K2_EXP == 5 || K2_EXP == 7) break;
        K2_ITEM_T0
        if (is_core && item < nA) {
            int ddx, ddy;
            k2_ring_pixel(item, ddx, ddy);
            const int X = x1 + ddx, Y = y1 + ddy;
            if (X >= 0 && X < size && Y >= 0 && Y < size) k2_wave_pixel<T>(X, Y, x1, y1, size, recA, recB, n_rays, order, start, map, alpha, sval[wv]);
        } else {
            int ddx, ddy;
            bool exists;
            if (arcs) { const int j = 64 * item + lane; exists = j < npo; k2_octant_pixel(xcd, exists ? j : 0, rB, ddx, ddy); }
            else      { const int j = pA + 64 * (item - nA) + lane; exists = j < n_pix; k2_ring_pixel(exists ? j : 0, ddx, ddy); }
            k2_lane_pixels<T>(ddx, ddy, exists, x1, y1, size, recA, recB, order, start, map, alpha, s_mixq, &s_nmix, K2_MIXQ);
        }
        K2_ITEM_T1(0, item)
    }
    K2_STAMP(2)
    // T3 (see above: its first two items were fetched before the zone)
    while (more0) {
        K2_ITEM_T0
        K2_FETCH(nxt, more1)
        if (cur.valid && K2_EXP != 9) {                               // (K2_EXP 9: the items are fetched and dropped; 10: every lane takes the lone-ray path; 11: no shortcut by xalone)
            const int smaj = ((cur.flags >> 2) & 3) - 1, ab = cur.b < 0 ? -cur.b : cur.b;
            if (K2_EXP == 10 || (K2_EXP != 11 && cur.x >= cur.xalone && ab != cur.x)) {
                // the ray shares no pixel from xalone on (diagonal pixels -- the quadrant's other class draws there too -- excepted)
                K2_EXP_STORE(map, cur.ptr, k2_blend(cur.pix, cur.x <= cur.lim2 ? TS_NO_OBSTACLE : k2_pixval_fast(cur.lim2, 0, recB[cur.ray], cur.x), alpha));
            } else {
                // Nearer the robot neighbouring rays are less than a pixel apart: which rays can draw the pixel is one contiguous
                // range of the slope-sorted table (the lane knows its pixel in its ray's own frame -- class, major offset x, minor
                // offset b: the range needs no classification).  Its own ray alone in it: blended at once.  A diagonal pixel or a
                // range with company goes on.
                int lo = 0, hi = 2;
                if (ab != cur.x) rs_range(start, (cur.flags & K2_F_MAJX) ? (smaj >= 0 ? 0 : 1) : (smaj >= 0 ? 2 : 3), cur.x, cur.b, 0.0f, lo, hi);
                const int v = cur.x <= cur.lim2 ? TS_NO_OBSTACLE : k2_pixval_fast(cur.lim2, cur.flags, recB[cur.ray], cur.x);
                if (hi - lo == 1) {
                    K2_EXP_STORE(map, cur.ptr, k2_blend(cur.pix, v, alpha));
                } else if (ab != cur.x && hi - lo <= K2_MAXHIT) {
                    // Company in the range: the OTHER rays of the range are tested -- this lane's own ray draws the pixel by
                    // construction.  A hit of a lower ray index ends the matter (that ray's lane owns the pixel); hits of higher
                    // indices are blended after this lane's value, in index order (at most K2_MAXHIT - 1 of them: kept sorted in
                    // registers).
                    int oidx[K2_MAXHIT - 1], oval[K2_MAXHIT - 1], no = 0;
                    bool owner = true;
                    for (int ci = lo; ci < hi && owner; ci++) {
                        const int ray = (int)order[ci];
                        if (ray == cur.ray) continue;
                        const k2_rayA e = recA[ray];
                        k2_cand c; c.dxc = e.dxc; c.sdyc = e.sdyc; c.lim2 = e.lim2; c.ray = ray;
                        if (!k2_hit<T>(c, cur.x, cur.b)) continue;
                        if (ray < cur.ray) { owner = false; break; }
                        const int ov = cur.x <= e.lim2 ? TS_NO_OBSTACLE : k2_pixval_fast(e.lim2, e.flags, recB[ray], cur.x);
                        int posn = 0;
#pragma unroll
                        for (int s2 = 0; s2 < K2_MAXHIT - 1; s2++) if (s2 < no && oidx[s2] < ray) posn++;
#pragma unroll
                        for (int s2 = K2_MAXHIT - 2; s2 >= 1; s2--) if (s2 > posn && s2 <= no) { oidx[s2] = oidx[s2 - 1]; oval[s2] = oval[s2 - 1]; }
#pragma unroll
                        for (int s2 = 0; s2 < K2_MAXHIT - 1; s2++) if (s2 == posn) { oidx[s2] = ray; oval[s2] = ov; }
                        no++;
                    }
                    if (owner) {
                        uint16_t pix = k2_blend(cur.pix, v, alpha);
#pragma unroll
                        for (int s2 = 0; s2 < K2_MAXHIT - 1; s2++) if (s2 < no) pix = k2_blend(pix, oval[s2], alpha);
                        K2_EXP_STORE(map, cur.ptr, pix);
                    }
                } else {
                    // a diagonal pixel, or more company than a lane orders in registers: all hits in ray order, by the lane of the lowest
                    const int a_s = smaj < 0 ? -cur.x : cur.x;
                    const int cdx = (cur.flags & K2_F_MAJX) ? a_s : cur.b, cdy = (cur.flags & K2_F_MAJX) ? cur.b : a_s;
                    bool owner;
                    const uint16_t pix = k2_lane_draw_ordered<T>(recA, recB, order, start, cdx, cdy, cur.pix, alpha, cur.ray, owner);
                    if (owner) K2_EXP_STORE(map, cur.ptr, pix);
                }
            }
        }
        cur = nxt; more0 = more1;
        K2_ITEM_T1(2, 0)
    }
#undef K2_FETCH
    K2_STAMP(3)
    if (BUILD && ride.on) { k3_ride rd = ride; if (sc.win_key) rd.d_pose = s_wpose; k2_ride_tail(rd, ride_cells, ride_ray, ride_cell, ride_r, ride_nw, ride_h, ride_nh, ride_v, ride_p); }
    if (sc.span) k2_row_spans<T>(recA, n_rays, x1, y1, size, n_pix_wgs, sc.span);    // (only while a host mirror is being kept: slamhip_cs_holemap_mirror_async)
    __syncthreads();
    K2_STAMP(4)
    // the zone pixels with hits inside a V that the one-lane items queued: the ordered way, four pixels to a wavefront
    if (core_count) {                                              // (a core workgroup's marked pixels: every ray by index, no sorted table)
        const int ncq = s_nmix;
        for (int item = wv; item < ncq; item += 16) {
            const int ptr = pos_s[2048 + item];
            const int py = ptr / size, px = ptr - py * size;
            k2_wave_pixel<T>(px, py, x1, y1, size, recA, recB, n_rays, order, start, map, alpha, sval[wv], true);
        }
        K2_STAMP(5)
        return;
    }
    const int nq = s_nmix < K2_MIXQ ? s_nmix : K2_MIXQ;
    for (int item = wv; item * 4 < nq; item += 16) {
        const int qi = item * 4 + (lane >> 4);
        const int ptr = s_mixq[qi < nq ? qi : 0];
        const int py = ptr / size, px = ptr - py * size;
        k2_wave_group<T>(px - x1, py - y1, qi < nq, 2, x1, y1, size, recA, recB, n_rays, order, start, map, alpha, sval[wv]);
    }
    K2_STAMP(5)
}

// ---- host side ----------------------------------------------------------------------------------------
int32_t cs_holemap_alloc(slamhip_cs *cs)
{
    SH_HIP(hipMalloc(&cs->d_k2_counters, sizeof(int) * 8));
    SH_HIP(hipMemsetAsync(cs->d_k2_counters, 0, sizeof(int) * 8, cs->ctx->stream));
    SH_HIP(hipMalloc(&cs->d_k2_start, sizeof(int) * (4 * K2_NBUCK + 1)));
    SH_HIP(hipMalloc(&cs->d_hole_dirty, sizeof(int) * 4));
    return cs_holemap_dirty_set(cs, true);
}

int32_t cs_holemap_dirty_set(slamhip_cs *cs, bool all)
{
    SH_HIP(hipMemsetD32Async((hipDeviceptr_t)cs->d_hole_dirty, all ? 0 : cs->hs, 2, cs->ctx->stream));
    SH_HIP(hipMemsetD32Async((hipDeviceptr_t)(cs->d_hole_dirty + 2), all ? cs->hs - 1 : -1, 2, cs->ctx->stream));
    return SLAMHIP_OK;
}

void cs_holemap_free(slamhip_cs *cs)
{
    (void)hipFree(cs->d_rays); (void)hipFree(cs->d_k2_cand); (void)hipFree(cs->d_k2_vprof); (void)hipFree(cs->d_k2_start);
    (void)hipFree(cs->d_k2_counters); (void)hipFree(cs->d_hole_dirty);
}

// with_obstacle: the ObstacleMap update of this scan rides on the launch (obstacle_dev.h); scans too large for the in-kernel
// tables take k2_prepare + the pixel kernel, and their ObstacleMap update its own launches
bool cs_holemap_one_launch(const slamhip_cs *cs)
{
    static const bool two_launch = getenv("SLAMHIP_K2_TWO_LAUNCHES") != nullptr;          // (tests: the large-scan path on ordinary scans)
    return cs->n_points > 0 && cs->n_points <= K2_LDS_RAYS && !two_launch;
}

int32_t cs_launch_holemap_update(slamhip_cs *cs, const float *d_pose, float4 h_pxcs, float4 h_pxcs_obst, float hole_width, int quality,
                                 bool with_obstacle, int max_hits, const cs_k2_winner *win)
{
    slamhip_ctx *ctx = cs->ctx;
    const int n = cs->n_points;
    if (n <= 0) return SLAMHIP_OK;
    SH_TRY(cs_flush_scan(cs));
    const bool build = cs_holemap_one_launch(cs);
    if (win && (!build || !d_pose)) SH_FAIL(SLAMHIP_ERR_STATE, "the key-decoding update needs the one-launch form and a pose buffer");
    k3_ride ride;
    memset(&ride, 0, sizeof(ride));
    if (with_obstacle && build) cs_obstacle_ride(cs, d_pose, h_pxcs_obst, max_hits, &ride);
    if (!build && n > cs->cap_rays) {
        if (cs->d_rays) (void)hipFree(cs->d_rays);
        if (cs->d_k2_cand) (void)hipFree(cs->d_k2_cand);
        if (cs->d_k2_vprof) (void)hipFree(cs->d_k2_vprof);
        cs->d_rays = nullptr; cs->d_k2_cand = nullptr; cs->d_k2_vprof = nullptr; cs->cap_rays = 0;
        const int cap = n + n / 4 + 64;
        SH_HIP(hipMalloc(&cs->d_rays, sizeof(k2_rayA) * (size_t)cap));               // the rays' records by index, parts A, B ...
        SH_HIP(hipMalloc(&cs->d_k2_vprof, sizeof(k2_rayB) * (size_t)cap));
        SH_HIP(hipMalloc(&cs->d_k2_cand, (sizeof(double) + sizeof(int)) * (size_t)cap));   // ... part C, and behind it the sorted table of ray indices
        cs->cap_rays = cap;
    }
    k2_scan sc;
    sc.pts = cs->d_pts; sc.scale = cs->hscale; sc.hole_width = hole_width; sc.d_pose = d_pose; sc.h_pxcs = h_pxcs;
    sc.total_out = (int *)cs->d_key + 6; sc.dirty = cs->d_hole_dirty;
    sc.span = cs->mirror_on ? cs->d_hole_span : nullptr;
    sc.win_key = nullptr; sc.win_offs = nullptr; sc.win_n_offs = 0; sc.win_bx = sc.win_by = sc.win_bth = 0.0f; sc.win_pose_out = nullptr; sc.win_mail = nullptr; sc.win_seq = 0;
    if (win) {
        sc.win_key = (const unsigned long long *)win->d_key; sc.win_offs = win->d_offs_flat; sc.win_n_offs = win->n_offs; sc.win_bx = win->bx; sc.win_by = win->by; sc.win_bth = win->bth;
        sc.win_pose_out = const_cast<float *>(d_pose); sc.win_mail = win->mail; sc.win_seq = win->seq;
    }
    static const int rb_env = getenv("SLAMHIP_K2_RB") ? atoi(getenv("SLAMHIP_K2_RB")) : 12, ncore_env = getenv("SLAMHIP_K2_NCORE") ? atoi(getenv("SLAMHIP_K2_NCORE")) : 1;
    sc.rb_num = rb_env < 1 ? 1 : rb_env;                           // (radius, per 1080 rays, from which a zone pixel is one lane's)
    sc.ncore = 0;                                                  // (set below, once the grid is known)
    static const int zone_env = getenv("SLAMHIP_K2_ZONE") ? atoi(getenv("SLAMHIP_K2_ZONE")) : 0;
    sc.zone = zone_env >= 1 ? zone_env : 0;                        // (developer override of the zone's radius; 0: by the ray count)
    {
        sh_timer t(ctx, SLAMHIP_K_CS_HOLEMAP);
        if (!build)
            hipLaunchKernelGGL(k2_prepare, dim3(1), dim3(1024), 0, ctx->stream, cs->d_pts, n, cs->hs, cs->hscale, d_pose, h_pxcs,
                               hole_width, (k2_rayA *)cs->d_rays, (k2_rayB *)cs->d_k2_vprof, (double *)cs->d_k2_cand, (int *)((double *)cs->d_k2_cand + cs->cap_rays),
                               cs->d_k2_start, cs->d_k2_counters, (int *)cs->d_key + 6, cs->d_hole_dirty);
        // One round of resident workgroups, one per CU (a second round would start when the first drains; with the tables in LDS
        // one workgroup per CU measured best).
        static const int grid_env = getenv("SLAMHIP_K2_GRID") ? atoi(getenv("SLAMHIP_K2_GRID")) : 0;
        const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
        // (two workgroups per CU -- the 1080-ray tables are 80.7 KB with the 16-bit bucket table, and amdgpu_waves_per_eu(8, 8) brings
        // the kernel under 80 SGPRs -- measured no faster: 23.5 against 23.2 us with one, and the register limit costs the
        // one-per-CU form a microsecond: 22.2 us without it)
        // (the rays beyond the zone are dealt to eight XCD sectors, sector s to the workgroups b with b % 8 == s: fewer than eight
        // workgroups would leave sectors undrawn -- the developer override is clamped)
        const int grid = grid_env > 0 ? (grid_env < 8 ? 8 : grid_env) : build ? cus : 2 * cus;
        // Arcs (k2_pixels): core workgroups with every ray + sector workgroups with their octant's rays.  Not while a host mirror's row
        // spans are kept (k2_row_spans walks every ray in every workgroup), not for grids that leave an XCD without a sector
        // workgroup, not for hole widths whose extension (:525-530) may turn a ray round or leave the range the margin is proved for.
        {
            const float hw_px = hole_width * cs->hscale * 0.5f;
            const bool hw_ok = hw_px >= 0.0f && hw_px < 8000.0f;
            const int nc = ncore_env <= 0 ? 0 : 1;                       // (one core workgroup per XCD: it draws its octant's central pixels by counting; SLAMHIP_K2_NCORE=0: no arcs)
            if (build && nc > 0 && !sc.span && hw_ok && cs->hs <= 16384 && grid % 8 == 0 && grid / 8 >= nc + 1) sc.ncore = nc;
        }
#define K2_PIXELS(B, T, A) {                                                                                                \
            static std::atomic<unsigned long long> attr_set{0};              /* one bit per device (the attribute is the device's) */   \
            if (!((attr_set.load(std::memory_order_acquire) >> (ctx->device & 63)) & 1ull)) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k2_pixels<B, T, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)k2_lds_bytes(B, B ? K2_LDS_RAYS : 0)); attr_set.fetch_or(1ull << (ctx->device & 63), std::memory_order_release); } \
            hipLaunchKernelGGL((k2_pixels<B, T, A>), dim3(grid), dim3(1024), k2_lds_bytes(B, n), ctx->stream, sc, (const k2_rayA *)cs->d_rays, \
                               (const k2_rayB *)cs->d_k2_vprof, (const double *)cs->d_k2_cand, (const int *)((const double *)cs->d_k2_cand + cs->cap_rays), \
                               n, (const int *)cs->d_k2_start, cs->d_k2_counters, \
                               cs->hs, cs->d_hole, quality, grid, ride); }
        if (build) { if (cs->hs <= 16384) { if (sc.ncore > 0) K2_PIXELS(true, int, true) else K2_PIXELS(true, int, false) } else K2_PIXELS(true, long long, false) }
        else       { if (cs->hs <= 16384) K2_PIXELS(false, int, false) else K2_PIXELS(false, long long, false) }
#undef K2_PIXELS
    }
    SH_HIP(hipGetLastError());
    if (with_obstacle) {
        if (build) cs_obstacle_ride_commit(cs, &ride, max_hits);
        else SH_TRY(cs_launch_obstacle_update(cs, d_pose, h_pxcs_obst, max_hits));
    }
#ifdef K2_TIMES
    {
        static thread_local int calls = 0;
        if (++calls == 12) {
            (void)hipStreamSynchronize(ctx->stream);
            std::vector<unsigned long long> h(512 * 8);
            (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_k2_times), sizeof(unsigned long long) * h.size());
            // (the symbol is zeroed after the dump: only the workgroups of this launch have stamps)
            unsigned long long t0 = ~0ull, t1 = 0;
            int nb = 0;
            for (int i = 0; i < 512; i++) if (h[i * 8] && h[i * 8 + 5] >= h[i * 8]) { nb++; t0 = std::min(t0, h[i * 8]); t1 = std::max(t1, h[i * 8 + 5]); }
            static const char *nm[5] = { "tables", "T1", "T3", "drain", "ticket" };
            double acc[5] = { 0 }, mx[5] = { 0 }, smax = 0;
            for (int i = 0; i < 512; i++) if (h[i * 8] && h[i * 8 + 5] >= h[i * 8]) {
                for (int k = 0; k < 5; k++) { const double d = (double)(h[i * 8 + k + 1] - h[i * 8 + k]) * 0.01; acc[k] += d; mx[k] = std::max(mx[k], d); }
                smax = std::max(smax, (double)(h[i * 8] - t0) * 0.01);
            }
            {
                double a6 = 0, a7 = 0, a1 = 0; int c = 0;
                for (int i = 0; i < 512; i++) if (h[i * 8] && h[i * 8 + 5] >= h[i * 8] && h[i * 8 + 6]) {
                    a6 += (double)(h[i * 8 + 6] - h[i * 8]) * 0.01; a7 += (double)(h[i * 8 + 7] - h[i * 8 + 6]) * 0.01; a1 += (double)(h[i * 8 + 1] - h[i * 8 + 7]) * 0.01; c++;
                }
                if (c) fprintf(stderr, "[k2 times] inside the table phase, mean: start .. first barrier %.2f | rays .. second barrier %.2f | prefix, scatter, third + fourth barrier %.2f\n", a6 / c, a7 / c, a1 / c);
            }
            {
                std::vector<unsigned long long> f(512 * 8);
                (void)hipMemcpyFromSymbol(f.data(), HIP_SYMBOL(g_k2_fine), sizeof(unsigned long long) * f.size());
                for (int role = 0; role < 2; role++) {
                    static const int nce = getenv("SLAMHIP_K2_NCORE") ? atoi(getenv("SLAMHIP_K2_NCORE")) : 1;
                    double a[6] = { 0 }; int c = 0;
                    for (int i = 0; i < 512; i++) if (h[i * 8] && h[i * 8 + 5] >= h[i * 8] && h[i * 8 + 6] && f[i * 8] && ((i < 8 * nce) == (role == 0))) {
                        a[0] += (double)(f[i * 8] - h[i * 8 + 6]) * 0.01; a[1] += (double)(f[i * 8 + 1] - f[i * 8]) * 0.01; a[2] += (double)(f[i * 8 + 2] - f[i * 8 + 1]) * 0.01;
                        a[3] += (double)(h[i * 8 + 7] - f[i * 8 + 2]) * 0.01; a[4] += (double)(f[i * 8 + 3] - h[i * 8 + 7]) * 0.01; a[5] += (double)(f[i * 8 + 4] - f[i * 8 + 3]) * 0.01; c++;
                    }
                    if (c) fprintf(stderr, "[k2 times] %s, first thread: points + selection %.2f | rays %.2f | wave reductions %.2f | barrier %.2f | bins read, scan, barrier %.2f | lists, barrier %.2f\n",
                                   role == 0 ? "core" : "sector", a[0] / c, a[1] / c, a[2] / c, a[3] / c, a[4] / c, a[5] / c);
                }
            }
            for (int role = 0; role < 2; role++) {       // core workgroups (the first `ncore` of every XCD: blocks 0 .. 8 * ncore - 1) and sector workgroups apart
                static const int ncore_env = getenv("SLAMHIP_K2_NCORE") ? atoi(getenv("SLAMHIP_K2_NCORE")) : 1;
                double a[8] = { 0 }, e5 = 0, m5 = 0; int c = 0;
                for (int i = 0; i < 512; i++) if (h[i * 8] && h[i * 8 + 5] >= h[i * 8] && h[i * 8 + 6] && ((i < 8 * ncore_env) == (role == 0))) {
                    a[0] += (double)(h[i * 8 + 6] - h[i * 8]) * 0.01; a[1] += (double)(h[i * 8 + 7] - h[i * 8 + 6]) * 0.01; a[2] += (double)(h[i * 8 + 1] - h[i * 8 + 7]) * 0.01;
                    a[3] += (double)(h[i * 8 + 2] - h[i * 8 + 1]) * 0.01; a[4] += (double)(h[i * 8 + 3] - h[i * 8 + 2]) * 0.01; a[5] += (double)(h[i * 8 + 4] - h[i * 8 + 3]) * 0.01; a[6] += (double)(h[i * 8 + 5] - h[i * 8 + 4]) * 0.01;
                    const double e = (double)(h[i * 8 + 5] - t0) * 0.01; e5 += e; m5 = std::max(m5, e); c++;
                }
                if (c) fprintf(stderr, "[k2 times] %s workgroups (%d): first barrier %.2f | selection + rays %.2f | prefix, lists, scatter %.2f | zone %.2f | steps %.2f | wait for the workgroup %.2f | queue %.2f | end at %.2f (max %.2f) us after the launch's first stamp\n",
                               role == 0 ? "core" : "sector", c, a[0] / c, a[1] / c, a[2] / c, a[3] / c, a[4] / c, a[5] / c, a[6] / c, e5 / c, m5);
            }
            fprintf(stderr, "[k2 times] %d workgroups, span %.2f us; first thread of each workgroup, mean (max):", nb, (double)(t1 - t0) * 0.01);
            for (int k = 0; k < 5; k++) fprintf(stderr, " %s %.2f (%.2f) |", nm[k], acc[k] / std::max(nb, 1), mx[k]);
            fprintf(stderr, " last workgroup starts at %.2f us\n", smax);
            {   // per XCD (workgroup b runs on XCD b % 8): when its workgroups reach the end of T1, of T3 and their ticket
                fprintf(stderr, "[k2 times] per XCD, mean (max) us from the launch's first stamp: ");
                for (int x = 0; x < 8; x++) {
                    double e2 = 0, e3 = 0, e5 = 0, m3 = 0, m5 = 0; int c = 0;
                    for (int i = x; i < 512; i += 8) if (h[i * 8] && h[i * 8 + 5] >= h[i * 8]) {
                        const double a2 = (double)(h[i * 8 + 2] - t0) * 0.01, a3 = (double)(h[i * 8 + 3] - t0) * 0.01, a5 = (double)(h[i * 8 + 5] - t0) * 0.01;
                        e2 += a2; e3 += a3; e5 += a5; m3 = std::max(m3, a3); m5 = std::max(m5, a5); c++;
                    }
                    if (c) fprintf(stderr, "[%d] T1 end %.1f, wave 0 leaves T3 %.1f (%.1f), ticket %.1f (%.1f) ", x, e2 / c, e3 / c, m3, e5 / c, m5);
                }
                fprintf(stderr, "\n");
            }
            std::vector<unsigned long long> sb(512 * 16 * 8);
            (void)hipMemcpyFromSymbol(sb.data(), HIP_SYMBOL(g_k2_sub), sizeof(unsigned long long) * sb.size());
            {   // the making of the rays, per wavefront, by how many rays it made
                double acc[5] = { 0 }; int cn[5] = { 0 };
                for (int w = 0; w < 512 * 16; w++) if (sb[w * 8 + 2]) { const int ns = (int)sb[w * 8 + 3], c = ns == 0 ? 0 : ns < 16 ? 1 : ns < 48 ? 2 : ns <= 64 ? 3 : 4; acc[c] += (double)sb[w * 8 + 2] * 0.01; cn[c]++; }
                fprintf(stderr, "[k2 times] a wavefront's ray loop, mean us by rays made: none %.2f (%d) | 1-15 %.2f (%d) | 16-47 %.2f (%d) | 48-64 %.2f (%d) | more %.2f (%d)\n",
                        acc[0] / std::max(cn[0], 1), cn[0], acc[1] / std::max(cn[1], 1), cn[1], acc[2] / std::max(cn[2], 1), cn[2], acc[3] / std::max(cn[3], 1), cn[3], acc[4] / std::max(cn[4], 1), cn[4]);
            }
            double tt[3] = { 0, 0, 0 }, cn[3] = { 0, 0, 0 }, wmax[3] = { 0, 0, 0 };
            struct top { double d; int kind, idx, wg, wv; };
            std::vector<top> tops;
            for (int w = 0; w < 512 * 16; w++) {
                for (int k = 0; k < 3; k++) { tt[k] += (double)sb[w * 8 + 2 * k] * 0.01; cn[k] += (double)sb[w * 8 + 2 * k + 1]; wmax[k] = std::max(wmax[k], (double)sb[w * 8 + 2 * k] * 0.01); }
                if (sb[w * 8 + 6]) tops.push_back({ (double)sb[w * 8 + 6] * 0.01, (int)(sb[w * 8 + 7] >> 16), (int)(sb[w * 8 + 7] & 65535), w / 16, w % 16 });
            }
            std::sort(tops.begin(), tops.end(), [](const top &a, const top &b) { return a.d > b.d; });
            fprintf(stderr, "[k2 times] per item, mean us (items; busiest wavefront's total): T1 %.2f (%.0f; %.2f) | T2 %.2f (%.0f; %.2f) | T3 %.2f (%.0f; %.2f)\n",
                    tt[0] / std::max(cn[0], 1.0), cn[0], wmax[0], tt[1] / std::max(cn[1], 1.0), cn[1], wmax[1], tt[2] / std::max(cn[2], 1.0), cn[2], wmax[2]);
            for (size_t i = 0; i < tops.size() && i < 12; i++) fprintf(stderr, "   longest items: %.2f us tier %d item %d (wg %d wave %d)\n", tops[i].d, tops[i].kind + 1, tops[i].idx, tops[i].wg, tops[i].wv);
        }
        if (calls == 11) {
            std::vector<unsigned long long> z(512 * 16 * 8, 0ull);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k2_times), z.data(), sizeof(unsigned long long) * 512 * 8);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k2_sub), z.data(), sizeof(unsigned long long) * z.size());
        }
    }
#endif
    return SLAMHIP_OK;
}
