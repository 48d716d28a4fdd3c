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
#define K2_ZONE 48                     // Chebyshev radius around the robot handled one wavefront per pixel
#define K2_MAXHIT 4                    // hits a lane-per-pixel thread orders in registers

// ray as the pixel kernels test it: clipped major length, signed clipped minor length (smin * dyc), the step beyond
// which pixval leaves TS_NO_OBSTACLE (:406), ray index (= blend order)
struct k2_cand { int dxc, sdyc, lim2, ray; };
// V-profile of a ray, by ray index: derrorv (:379/:386), incv (:398), lim2 = dx - 2*derrorv, lim1 = dx - derrorv (:406,:408)
struct k2_vprof { int derrorv, incv, lim2, lim1; };
// ray by index, for the pixels that scan all rays: flags = valid | major_x << 1 | (smaj + 1) << 2
struct k2_byidx { int dxc, sdyc, lim2, flags; };

struct cs_ray {
    int valid;
    int ptr0;                 // y1*Size + x1                         (:401)
    int x1, y1;
    int dx;                   // unclipped major length after swap     (:368,:383)
    int dxc, dyc;             // clipped major / minor length          (:370-371,:384)
    int incmaj, incmin;       // ptr increments after swap             (:372-373,:385)
    int major_x;              // 1: major axis is x
    int smaj, smin;           // coordinate signs along major / minor
    int derrorv, incv, incerrorv, sincv;   // :379/:386, :398, :399, :374
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
    // for some inputs -- found by tests/fuzz_parity.py as a ray end one pixel out.  The binary64 square root of a binary32
    // value, rounded once more to binary32, is the correctly rounded binary32 root: 53 >= 2 * 24 + 2.)
    const float dist = (float)sqrt((double)(x2p * x2p + y2p * y2p));       // :524
    const float add = __fdiv_rn(__fdiv_rn(hole_width * scale, 2.0f), dist); // :525
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
            r.sincv = sh_sign(TS_OBSTACLE - TS_NO_OBSTACLE);               // :374
            r.incv = (TS_OBSTACLE - TS_NO_OBSTACLE) / derrorv;             // :398
            r.incerrorv = sh_wsub(TS_OBSTACLE - TS_NO_OBSTACLE, sh_wmul(derrorv, r.incv));   // :399
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
__device__ static __forceinline__ int k2_pixval_closed(const k2_vprof p, int x)
{
    if (x <= p.lim2) return TS_NO_OBSTACLE;
    const int d = p.derrorv;
    const int incerrorv = sh_wsub(TS_OBSTACLE - TS_NO_OBSTACLE, sh_wmul(d, p.incv));   // :399, in (-d, 0]
    if (d > (1 << 24)) {
        // absurd hole widths (half-width beyond 16M pixels): the closed form's intermediates could leave int32, where the
        // reference's unchecked arithmetic wraps -- walk the recurrence (:406-428) literally instead
        int pixval = TS_NO_OBSTACLE, errorv = d / 2;                       // :402,:397
        for (int xi = p.lim2 < 0 ? 0 : p.lim2 + 1; xi <= x; xi++) {
            if (xi <= p.lim1) {                                            // :408
                pixval = sh_wadd(pixval, p.incv);
                errorv = sh_wadd(errorv, incerrorv);
                if (errorv > d) { pixval = sh_wadd(pixval, -1); errorv = sh_wsub(errorv, d); }
            } else {
                pixval = sh_wsub(pixval, p.incv);
                errorv = sh_wsub(errorv, incerrorv);
                if (errorv < 0) { pixval = sh_wsub(pixval, -1); errorv = sh_wadd(errorv, d); }
            }
        }
        return pixval;
    }
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

// counters: [0] R = longest clipped major length, [1] conflict pixels, [2] blended pixels (every step x = 0..dxc of
// a valid ray blends exactly one pixel, :404,:431), [3] x1, [4] y1, [5] robot inside the map
__global__ void __launch_bounds__(1024)
k2_prepare(const float2 *__restrict__ pts, int n, int size, float scale, const float *d_pose, float4 h_pxcs, float hole_width,
           k2_byidx *__restrict__ byidx, k2_cand *__restrict__ cand, k2_vprof *__restrict__ vprof,
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
    k2_byidx keep[2];                                              // a thread's first two rays stay in registers for the second pass (scans of up to 2048 rays)
    keep[0].flags = 0; keep[1].flags = 0;
    for (int i = t, it = 0; i < n; i += 1024, it++) {
        const cs_ray r = k2_make_ray(pts[i], size, q, scale, hole_width);
        k2_byidx e; e.dxc = r.dxc; e.sdyc = r.smin * r.dyc; e.lim2 = r.lim2;
        e.flags = (r.valid ? 1 : 0) | (r.major_x ? 2 : 0) | ((r.smaj + 1) << 2);
        byidx[i] = e;
        if (it == 0) keep[0] = e; else if (it == 1) keep[1] = e;
        if (r.valid) {
            k2_vprof vp; vp.derrorv = r.derrorv; vp.incv = r.incv; vp.lim2 = r.lim2; vp.lim1 = r.lim1;
            vprof[i] = vp;
            const int cls = r.major_x ? (r.smaj >= 0 ? 0 : 1) : (r.smaj >= 0 ? 2 : 3);
            const float tt = r.dxc > 0 ? (float)e.sdyc / (float)r.dxc : 0.0f;
            atomicAdd(&hist[cls * K2_NBUCK + rs_bucket(tt)], 1);
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
    for (int i = t, it = 0; i < n; i += 1024, it++) {
        const k2_byidx e = it == 0 ? keep[0] : it == 1 ? keep[1] : byidx[i];      // (its own store: no other thread wrote byidx[i])
        if (e.flags & 1) {
            const int smaj = ((e.flags >> 2) & 3) - 1;
            const int cls = (e.flags & 2) ? (smaj >= 0 ? 0 : 1) : (smaj >= 0 ? 2 : 3);
            const float tt = e.dxc > 0 ? (float)e.sdyc / (float)e.dxc : 0.0f;
            const int pos = atomicAdd(&hist[cls * K2_NBUCK + rs_bucket(tt)], 1);
            k2_cand c; c.dxc = e.dxc; c.sdyc = e.sdyc; c.lim2 = e.lim2; c.ray = i;
            cand[pos] = c;
        }
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
// (byidx / cand / vprof: LDS when the kernel made the scan's tables itself, else global; vprof goes by ray index)
template <typename T, typename CT, typename BT, typename VT, typename ST>
__device__ static __forceinline__ void k2_wave_pixel(int X, int Y, int x1, int y1, int size, BT byidx,
                                            VT vprof, int n_rays, CT cand, const ST *start,
                                            uint16_t *__restrict__ map, int alpha, int *sval)
{
    const int lane = threadIdx.x & 63;
    const int ptr = Y * size + X;
    const int dx = X - x1, dy = Y - y1;
    int cls[2], a[2], b[2], lo[2] = { 0, 0 }, hi[2] = { 0, 0 };
    const int ncls = rs_classes(dx, dy, cls, a, b);
    int nc = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) if (k < ncls) { rs_range(start, cls[k], a[k], b[k], 0.0f, lo[k], hi[k]); nc += hi[k] - lo[k]; }
    uint16_t pix = map[ptr];
    bool stable = false;
    int last_v = 0;
    if (ncls == 0 || nc > 64) {
        k2_byidx e_next = byidx[lane < n_rays ? lane : 0];             // (the table is read one iteration ahead)
        for (int base = 0; base < n_rays; base += 64) {
            const int i = base + lane;
            bool hit = false;
            int v = 0;
            const k2_byidx e = e_next;
            e_next = byidx[i + 64 < n_rays ? i + 64 : 0];
            if (i < n_rays) {
                if (e.flags & 1) {
                    const int smaj = ((e.flags >> 2) & 3) - 1;
                    const int aa = (e.flags & 2) ? dx : dy, bb = (e.flags & 2) ? dy : dx;
                    int x = -1;
                    if (smaj != 0) x = aa * smaj; else if (aa == 0) x = 0;
                    k2_cand c; c.dxc = e.dxc; c.sdyc = e.sdyc; c.lim2 = e.lim2; c.ray = i;
                    if (x == 0 ? bb == 0 : (x > 0 && k2_hit<T>(c, x, bb))) {
                        hit = true;
                        v = x <= e.lim2 ? TS_NO_OBSTACLE : k2_pixval_closed(vprof[i], x);
                    }
                }
            }
            unsigned long long mask = __ballot(hit);
            while (mask) {
                // the leading run of hits with one value (near the robot nearly every ray carries TS_NO_OBSTACLE): the
                // blend of a run converges -- once it no longer changes the pixel the rest of the run cannot either
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
    } else if (nc > 0) {
        int ci = -1, kk = 0;
        if (lane < hi[0] - lo[0]) { ci = lo[0] + lane; kk = 0; }
        else if (ncls > 1 && lane - (hi[0] - lo[0]) < hi[1] - lo[1]) { ci = lo[1] + lane - (hi[0] - lo[0]); kk = 1; }
        bool hit = false;
        int idx = 0x7fffffff, v = 0;
        if (ci >= 0) {
            const k2_cand c = cand[ci];
            const int aa = kk ? a[1] : a[0], bb = kk ? b[1] : b[0];
            if (k2_hit<T>(c, aa, bb)) { hit = true; idx = c.ray; v = aa <= c.lim2 ? TS_NO_OBSTACLE : k2_pixval_closed(vprof[c.ray], aa); }
        }
        const unsigned long long mask = __ballot(hit);
        if (mask) {
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
//   T1  r < K2_ZONE (48)   pixel-centric, wavefronts: tens to a thousand rays cross a pixel near the robot, a handful at r = 47.
//                          Pixels are numbered from the centre outwards (k2_ring_pixel); a wavefront takes one pixel (r < rB), two
//                          (rB <= r < rC: 32 lanes test the candidate rays of each) or four (16 lanes each): lanes test the candidates,
//                          hits are rank-sorted by ray index through LDS and blended in that order by the group's first lane.
//   T3  r >= K2_ZONE       one lane per (ray, step): the lane computes its pixel from the closed form of the walk and asks, like a
//                          pixel-centric lane would, which rays can draw that pixel -- one contiguous range of the slope-sorted
//                          table.  Out here rays are more than a pixel apart: nearly always the range holds the lane's own ray and
//                          nothing else, and the pixel is blended at once.  Otherwise (and on the diagonals, where the two classes
//                          of a quadrant meet) the candidates are tested; the lane of the LOWEST ray index among the hits owns the
//                          pixel and blends all hits in ray order, the other hitting rays' lanes drop it.  The lookup is a function
//                          of the pixel alone, so every lane that lands on a pixel sees the same hit list and exactly one owns it.
// Pixels with more hits than a T3 lane orders go to the conflict list (drawn by the last workgroup, one wavefront per pixel).
// The bucket table, the sorted ray table, the V-profiles (in table order) and the rays by index live in LDS: a pixel's
// lookup is a chain of dependent small reads (bucket bounds -> candidates -> V-profile -> map), which global-memory latency
// would dominate.
#ifndef K2_LDS_RAYS
#define K2_LDS_RAYS 2400               // largest scan whose tables fit the LDS: 2 x 16.4 KB of buckets + 48 B per ray + the kernel's static 4 KB <= 160 KB
#endif
#ifdef K2_TIMES
// developer instrumentation (build with SLAMHIP_K2_TIMES=1): 100 MHz wall-clock stamps per workgroup and phase
__device__ unsigned long long g_k2_times[512 * 8];
#define K2_STAMP(k) { if (threadIdx.x == 0 && blockIdx.x < 512) g_k2_times[blockIdx.x * 8 + (k)] = wall_clock64(); }
__device__ unsigned long long g_k2_sub[512 * 16 * 8];   // per wavefront: [0] T1 time [1] T1 items [2] - [3] - [4] T3 time [5] T3 items [6] longest item [7] its kind * 65536 + index
#define K2_ITEM_T0 const unsigned long long it0_ = wall_clock64();
#define K2_ITEM_T1(kind, idx) { if ((threadIdx.x & 63) == 0 && blockIdx.x < 512) { unsigned long long *p_ = g_k2_sub + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8; \
        const unsigned long long d_ = wall_clock64() - it0_; p_[2 * (kind)] += d_; p_[2 * (kind) + 1] += 1; if (d_ > p_[6]) { p_[6] = d_; p_[7] = (unsigned long long)(kind) * 65536ull + (unsigned long long)((idx) & 65535); } } }
#else
#define K2_STAMP(k) {}
#define K2_ITEM_T0
#define K2_ITEM_T1(kind, idx)
#endif

// Which rays draw the pixel at offset (dx, dy) from the robot?  Up to H hits, kept sorted by ray index (compile-time
// subscripts: the lists stay in registers); `min_ray` = the lowest hitting ray index, also when the list overflowed.
template <typename T, int H, typename CT, typename VT, typename ST>
__device__ static __forceinline__ void k2_lookup(CT cand, VT vprof, const ST *start, int dx, int dy,
                                        int (&hidx)[H], int (&hval)[H], int &nh, bool &overflow, int &min_ray)
{
    int cls[2], a[2], b[2];
    const int ncls = rs_classes(dx, dy, cls, a, b);
    nh = 0; overflow = false; min_ray = 0x7fffffff;
    for (int k = 0; k < ncls; k++) {
        int lo, hi;
        rs_range(start, cls[k], a[k], b[k], 0.0f, lo, hi);
        for (int ci = lo; ci < hi; ci++) {
            const k2_cand c = cand[ci];
            if (!k2_hit<T>(c, a[k], b[k])) continue;
            min_ray = c.ray < min_ray ? c.ray : min_ray;
            if (nh == H) { overflow = true; continue; }            // (keep scanning: the owner is the lowest index of ALL hits)
            const int v = a[k] <= c.lim2 ? TS_NO_OBSTACLE : k2_pixval_closed(vprof[c.ray], a[k]);
            int posn = 0;
#pragma unroll
            for (int s = 0; s < H; s++) if (s < nh && hidx[s] < c.ray) posn++;
#pragma unroll
            for (int s = H - 1; s >= 1; s--) if (s > posn && s <= nh) { hidx[s] = hidx[s - 1]; hval[s] = hval[s - 1]; }
#pragma unroll
            for (int s = 0; s < H; s++) if (s == posn) { hidx[s] = c.ray; hval[s] = v; }
            nh++;
        }
    }
}

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

// One wavefront draws G = 1 << lg pixels (numbers pix0 .. pix0 + G - 1 of the zone), W = 64 / G lanes each.  A pixel with more
// candidates than its lanes -- and every G = 1 item -- goes down the one-pixel path, pixel after pixel: ONE call site for it (the
// kernel's code is executed once or twice per wavefront, from a cold instruction cache: its size is latency; nine inlined copies
// of the one-pixel path made a 47 KB kernel that ran 10 us slower than the 20 KB one).
template <typename T, typename CT, typename BT, typename VT, typename ST>
__device__ static __forceinline__ void k2_wave_group(int pix0, int n_pix, int lg, int x1, int y1, int size, BT byidx,
                                                     VT vps, int n_rays, CT cand, const ST *start, uint16_t *__restrict__ map, int alpha, int *sval)
{
    const int W = 64 >> lg, G = 1 << lg;
    const int lane = threadIdx.x & 63, g = lane >> (6 - lg), l = lane & (W - 1);
    int ddx, ddy;
    k2_ring_pixel(pix0 + g, ddx, ddy);
    const int X = x1 + ddx, Y = y1 + ddy;
    const bool valid = pix0 + g < n_pix && X >= 0 && X < size && Y >= 0 && Y < size;
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
            if (vg) k2_wave_pixel<T>(Xg, Yg, x1, y1, size, byidx, vps, n_rays, cand, start, map, alpha, sval);
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
        const k2_cand c = cand[ci];
        const int aa = kk ? a[1] : a[0], bb = kk ? b[1] : b[0];
        if (k2_hit<T>(c, aa, bb)) { hit = true; idx = c.ray; v = aa <= c.lim2 ? TS_NO_OBSTACLE : k2_pixval_closed(vps[c.ray], aa); }
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

// dynamic LDS of the pixel kernel.  !BUILD: the bucket table (int).  BUILD (the kernel makes the scan's tables itself): the
// histogram / running positions of the counting sort (int: LDS atomics), the bucket table as unsigned short (a scan has at most
// K2_LDS_RAYS rays), the sorted ray table, the V-profiles and the rays by index (16 bytes per ray each) -- 80.7 KB with the
// kernel's static 4.3 KB at 1080 rays.
#define K2_LDS_FIXED ((4 * K2_NBUCK + 4) * 4)
#define K2_LDS_START16 ((4 * K2_NBUCK + 8) * 2)
static inline size_t k2_lds_bytes(bool build, int n_rays) { return build ? (size_t)4 * K2_NBUCK * 4 + K2_LDS_START16 + (size_t)48 * (size_t)((n_rays + 3) & ~3) : (size_t)K2_LDS_FIXED; }

// a T3 work item as a lane holds it between its fetch (the pixel's load is issued there) and its turn
struct k2_t3 { int ptr, dx, dy, ray, lim2; uint16_t pix; };

// what the kernel needs of the scan when it makes the tables itself
struct k2_scan { const float2 *pts; float scale, hole_width; const float *d_pose; float4 h_pxcs; int *total_out; int *dirty; int rb_num, rc_num; int2 *span;
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
__device__ static __noinline__ void k2_row_spans(const k2_byidx *byidx, int n_rays, int x1, int y1, int size, int n_pix_wgs, int2 *__restrict__ span)
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
            const k2_byidx e = byidx[i];
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

template <bool BUILD, typename T>
__global__ void __launch_bounds__(1024)
k2_pixels(const k2_scan sc, const k2_byidx *__restrict__ byidx_g, const k2_vprof *__restrict__ vprof_g,
          const k2_cand *__restrict__ cand_g, int n_rays,
          const int *__restrict__ start_g, int *__restrict__ counters, int size, uint16_t *__restrict__ map, int alpha,
          int *__restrict__ conflict_pix, int cap_conflict, int n_pix_wgs, const k3_ride ride)
{
    extern __shared__ __attribute__((aligned(16))) char k2_smem[];
    typedef typename std::conditional<BUILD, unsigned short, int>::type start_t;
    const int n4 = (n_rays + 3) & ~3;
    int *pos_s = (int *)k2_smem;                                   // BUILD: histogram, then the buckets' running positions
    start_t *start = (start_t *)(k2_smem + (BUILD ? 4 * K2_NBUCK * 4 : 0));
    k2_cand *cand_s = (k2_cand *)(k2_smem + 4 * K2_NBUCK * 4 + K2_LDS_START16);
    k2_vprof *vprof_s = (k2_vprof *)(cand_s + (BUILD ? n4 : 0));
    k2_byidx *byidx_s = (k2_byidx *)(vprof_s + (BUILD ? n4 : 0));
    __shared__ __attribute__((aligned(16))) int sval[16][64];
    __shared__ int s_last, s_nextA, s_nextB, s_R, s_total, wsum[16];
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
    int R, x1, y1, n_valid;
    if (BUILD) {
        const int t = threadIdx.x, lane_ = t & 63, wid = t >> 6;
        constexpr int RPT = (K2_LDS_RAYS + 1023) / 1024;            // rays per thread
        float2 p_next = make_float2(0.f, 0.f);
        if (t < n_rays) p_next = sc.pts[t];                         // (a thread's next point is requested one iteration ahead)
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
        if (t == 0) { s_nextA = 0; s_nextB = 0; s_R = 0; s_total = 0; }
        __syncthreads();
        K2_STAMP(6)
        int bkt[RPT];                                               // a ray's bucket (class * 1024 + slope bucket), -1: not valid
        int my_R = 0, my_total = 0;
#pragma unroll
        for (int k = 0; k < RPT; k++) bkt[k] = -1;
#pragma unroll 1
        for (int it = 0; it * 1024 < n_rays; it++) {                // (rolled: k2_make_ray is kilobytes of code, fetched once per launch)
            const int i = t + it * 1024;
            int bb = -1;
            const float2 p = p_next;
            if (i + 1024 < n_rays) p_next = sc.pts[i + 1024];
            if (i < n_rays) {
                const cs_ray r = k2_make_ray(p, size, q, sc.scale, sc.hole_width);
                k2_byidx ee; ee.dxc = r.dxc; ee.sdyc = r.smin * r.dyc; ee.lim2 = r.lim2;
                ee.flags = (r.valid ? 1 : 0) | (r.major_x ? 2 : 0) | ((r.smaj + 1) << 2);
                byidx_s[i] = ee;
                if (r.valid) {
                    k2_vprof vv; vv.derrorv = r.derrorv; vv.incv = r.incv; vv.lim2 = r.lim2; vv.lim1 = r.lim1;
                    vprof_s[i] = vv;
                    const float tt = r.dxc > 0 ? (float)ee.sdyc / (float)r.dxc : 0.0f;
                    bb = (r.major_x ? (r.smaj >= 0 ? 0 : 1) : (r.smaj >= 0 ? 2 : 3)) * K2_NBUCK + rs_bucket(tt);
                    atomicAdd(&pos_s[bb], 1);
                    my_R = max(my_R, r.dxc);
                    my_total += r.dxc + 1;
                }
            }
#pragma unroll
            for (int k = 0; k < RPT; k++) if (k == it) bkt[k] = bb;
        }
        for (int off = 32; off > 0; off >>= 1) {                   // one LDS atomic per wave, not per ray (same address)
            my_R = max(my_R, __shfl_down(my_R, off, 64));
            my_total += __shfl_down(my_total, off, 64);
        }
        if (lane_ == 0) { atomicMax(&s_R, my_R); atomicAdd(&s_total, my_total); }
        __syncthreads();
        K2_STAMP(7)
        {   // exclusive prefix over the 4096 bins: 4 consecutive bins per thread
            int v[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = pos_s[4 * t + k]; sum += v[k]; }
            int incl = sum;
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (lane_ >= off) incl += o;
            }
            if (lane_ == 63) wsum[wid] = incl;
            __syncthreads();
            int base = incl - sum;
            for (int w = 0; w < wid; w++) base += wsum[w];
#pragma unroll
            for (int k = 0; k < 4; k++) { start[4 * t + k] = (start_t)base; pos_s[4 * t + k] = base; base += v[k]; }     // (each thread its own four bins)
            if (t == 1023) start[4 * K2_NBUCK] = (start_t)base;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < RPT; it++) {
            const int i = t + it * 1024;
            if (i < n_rays && bkt[it] >= 0) {
                const k2_byidx ee = byidx_s[i];                    // (this thread's own store)
                const int pos = atomicAdd(&pos_s[bkt[it]], 1);
                k2_cand c; c.dxc = ee.dxc; c.sdyc = ee.sdyc; c.lim2 = ee.lim2; c.ray = i;
                cand_s[pos] = c;
            }
        }
        R = s_R; x1 = sh_f2i(q.x); y1 = sh_f2i(q.y);
        if (blockIdx.x == 0 && t == 0) {                           // what the host reads: reach, blended pixels, the robot's pixel
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
        n_valid = (int)start[4 * K2_NBUCK];
        if (x1 < 0 || x1 >= size || y1 < 0 || y1 >= size) {         // robot outside the map: nothing is drawn (:509-512)
            if (ride.on) { k3_ride rd = ride; if (sc.win_key) rd.d_pose = s_wpose; k2_ride_tail(rd, ride_cells, ride_ray, ride_cell, ride_r, ride_nw, ride_h, ride_nh, ride_v, ride_p); }   // (the ObstacleMap has its own test, at its own scale :557-560)
            return;
        }
    } else {
        R = counters[0]; x1 = counters[3]; y1 = counters[4];
        if (x1 < 0 || x1 >= size || y1 < 0 || y1 >= size) return;   // robot outside the map: nothing is drawn (:509-512)
        if (threadIdx.x == 0) { s_nextA = 0; s_nextB = 0; }
        for (int i = threadIdx.x; i <= 4 * K2_NBUCK; i += 1024) start[i] = (start_t)start_g[i];
        __syncthreads();
        n_valid = (int)start[4 * K2_NBUCK];
    }
    K2_STAMP(1)
    const k2_cand *cand = BUILD ? cand_s : cand_g;
    const k2_vprof *vps = BUILD ? vprof_s : vprof_g;
    const k2_byidx *byidx = BUILD ? byidx_s : byidx_g;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Work items are dealt to the workgroups round-robin and inside a workgroup to whichever wavefront is free (an LDS
    // counter): an item costs what its pixels' hit lists cost, and a workgroup is only as fast as its slowest wavefront.
    // T1: the zone, from the centre outwards (the closer to the robot, the more rays cross a pixel: the longest items start
    // first).  About n / (2 pi r) rays cross a pixel at distance r, and a pixel has a fifth more candidates than that: two pixels
    // per wavefront from rB on (~17 candidates for 32 lanes), four from rC on (~7 for 16 lanes) -- a pixel with more candidates
    // than its lanes sends the whole item down the one-pixel path, which costs as much as the pixels it holds.
    const int Z = K2_ZONE - 1 < R ? K2_ZONE - 1 : R, n_pix = (2 * Z + 1) * (2 * Z + 1);
    int rB = (sc.rb_num * n_valid + 1079) / 1080, rC = (sc.rc_num * n_valid + 1079) / 1080;
    rB = rB < 1 ? 1 : rB > K2_ZONE ? K2_ZONE : rB; rC = rC < rB ? rB : rC > K2_ZONE ? K2_ZONE : rC;
    const int pA = min((2 * rB - 1) * (2 * rB - 1), n_pix), pB = min((2 * rC - 1) * (2 * rC - 1), n_pix);
    const int nA = pA, nB = (pB - pA + 1) / 2, nC = (n_pix - pB + 3) / 4;
    for (;;) {
        int k;
        SH_WAVE_FETCH(k, atomicAdd(&s_nextA, 1))
        const int item = blockIdx.x + k * n_pix_wgs;
        if (item >= nA + nB + nC) break;
        K2_ITEM_T0
        int pix0 = item, lim = n_pix, lg = 0;
        if (item >= nA + nB) { pix0 = pB + 4 * (item - nA - nB); lg = 2; }
        else if (item >= nA) { pix0 = pA + 2 * (item - nA); lim = pB; lg = 1; }
        k2_wave_group<T>(pix0, lim, lg, x1, y1, size, byidx, vps, n_rays, cand, start, map, alpha, sval[wv]);
        K2_ITEM_T1(0, item)
    }
    K2_STAMP(2)
    // T3: one lane per (ray, step) beyond the zone, ray = an entry of the sorted table, steps in blocks of 64.  Software
    // pipeline: an item's pixel is requested when the item is fetched, one iteration before its turn -- the map sits in HBM /
    // Infinity Cache, a microsecond away.
    // (Measured and rejected, round 3: sectors of equal WORK instead of equal counts -- a prefix sum over the rays' blocks of 64
    // steps in the table phase, bounds where it passes k/8 of the total, a sector's blocks ending with its own longest ray.  On the
    // benchmark scan the equal-count sectors hold 574 .. 1620 non-empty blocks, but their XCDs finish within 1.5 us of each other
    // -- a block near the robot, where rays lie a pixel apart and pixels have several candidates, costs several times one far out --
    // and the equal-work form was no better balanced and paid 3 us in the table phase.)
    // Rays are dealt BY INDEX (a scan's rays come in order of their angle: neighbours in index are neighbours in direction), and to
    // the XCDs by sector -- XCD s (workgroup b runs on XCD b % 8) draws the s-th eighth of the scan: a ray's pixels share their
    // 128-byte lines with its neighbours' (at r = 600 px adjacent rays are 3.5 px apart), and a line should meet one L2.
    const int nblk = R >= K2_ZONE ? (R - K2_ZONE) / 64 + 1 : 0;      // steps K2_ZONE .. R
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_in_xcd = (n_pix_wgs - xcd + 7) >> 3;
    const int c0 = (int)(((long long)n_rays * xcd) >> 3), n_sec = (int)(((long long)n_rays * (xcd + 1)) >> 3) - c0;
    const int n_t3 = nblk * n_sec;
    const float rcp_nv = __builtin_amdgcn_rcpf((float)(n_sec > 0 ? n_sec : 1));
#define K2_FETCH(it, more_)                                                                              \
    {                                                                                                   \
        int k_;                                                                                         \
        SH_WAVE_FETCH(k_, atomicAdd(&s_nextB, 1))                                                       \
        const int item_ = wg_in_xcd + k_ * wgs_in_xcd;                                                  \
        (it).ptr = -1;                                                                                  \
        more_ = item_ < n_t3;                                                                           \
        if (more_) {                                                                                    \
            int blk_ = (int)((float)item_ * rcp_nv);       /* item / n_sec (item < 2^24: settled exactly below) */ \
            int ri_ = item_ - blk_ * n_sec;                                                             \
            if (ri_ < 0) { blk_--; ri_ += n_sec; } else if (ri_ >= n_sec) { blk_++; ri_ -= n_sec; }     \
            ri_ += c0;                                                                                  \
            const k2_byidx me_ = byidx[ri_];               /* (uniform: an LDS broadcast) */             \
            const int x_ = K2_ZONE + blk_ * 64 + lane;                                                  \
            if ((me_.flags & 1) && x_ <= me_.dxc) {                                                     \
                const int smaj_ = ((me_.flags >> 2) & 3) - 1;                                           \
                const int dyc_ = me_.sdyc < 0 ? -me_.sdyc : me_.sdyc;                                   \
                const T N_ = (T)2 * dyc_ * x_ - me_.dxc, D_ = (T)2 * me_.dxc;                           \
                int m_ = 0;                                                                             \
                if (N_ > 0) {      /* m(x) = min(x, ceil(N / D)), the closed form of the error recurrence (:394-396, :433-441) */ \
                    T q_;                                                                               \
                    if (sizeof(T) == 4) {  /* N < 2^29, D < 2^16: the float estimate of floor(N / D) is within one; one multiply settles it */ \
                        q_ = (T)((float)N_ * __builtin_amdgcn_rcpf((float)D_));                         \
                        T r_ = N_ - q_ * D_;                                                            \
                        if (r_ < 0) { q_--; r_ += D_; } else if (r_ >= D_) { q_++; r_ -= D_; }          \
                        q_ += r_ > 0 ? 1 : 0;                      /* ceil */                            \
                    } else q_ = (N_ + D_ - 1) / D_;                                                     \
                    m_ = q_ < (T)x_ ? (int)q_ : x_;                                                     \
                }                                                                                       \
                const int b_ = me_.sdyc < 0 ? -m_ : m_, a_ = smaj_ < 0 ? -x_ : x_;                      \
                (it).dx = (me_.flags & 2) ? a_ : b_; (it).dy = (me_.flags & 2) ? b_ : a_;               \
                (it).ray = ri_; (it).lim2 = me_.lim2;                                                   \
                (it).ptr = (y1 + (it).dy) * size + (x1 + (it).dx);         /* (step pixels of a clipped ray lie inside the map) */ \
                (it).pix = map[(it).ptr];                                                               \
            }                                                                                           \
        }                                                                                               \
    }
    k2_t3 cur, nxt;
    cur.ptr = -1; cur.dx = cur.dy = cur.ray = cur.lim2 = 0; cur.pix = 0; nxt = cur;
    bool more = false;
    if (n_t3 > 0) K2_FETCH(cur, more)
    while (more) {
        K2_ITEM_T0
        K2_FETCH(nxt, more)
        if (cur.ptr >= 0) {
            // Out here rays are more than a pixel apart: nearly every pixel's candidate range holds its own ray and nothing else
            // -- then it is blended at once (no hit test, no ordering).  A diagonal pixel (the quadrant's other class draws there
            // too) or a range with company goes through the full lookup.
            const int adx = cur.dx < 0 ? -cur.dx : cur.dx, ady = cur.dy < 0 ? -cur.dy : cur.dy;
            int lo = 0, hi = 2;
            if (adx != ady) {
                const bool xm = adx > ady;
                rs_range(start, xm ? (cur.dx > 0 ? 0 : 1) : (cur.dy > 0 ? 2 : 3), xm ? adx : ady, xm ? cur.dy : cur.dx, 0.0f, lo, hi);
            }
            if (hi - lo == 1) {
                const int a = adx > ady ? adx : ady;
                const int v = a <= cur.lim2 ? TS_NO_OBSTACLE : k2_pixval_closed(vps[cur.ray], a);
                map[cur.ptr] = k2_blend(cur.pix, v, alpha);
            } else {
                int hidx[K2_MAXHIT], hval[K2_MAXHIT], nh, min_ray;
                bool overflow;
                k2_lookup<T, K2_MAXHIT>(cand, vps, start, cur.dx, cur.dy, hidx, hval, nh, overflow, min_ray);
                if (min_ray == cur.ray) {                          // the owner
                    if (overflow) {
                        const int slot = __hip_atomic_fetch_add(&counters[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (slot < cap_conflict) __hip_atomic_store(&conflict_pix[slot], cur.ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        uint16_t pix = cur.pix;
#pragma unroll
                        for (int s2 = 0; s2 < K2_MAXHIT; s2++) if (s2 < nh) pix = k2_blend(pix, hval[s2], alpha);
                        map[cur.ptr] = pix;
                    }
                }
            }
        }
        cur = nxt;
        K2_ITEM_T1(2, 0)
    }
#undef K2_FETCH
    // pixels with more hits than a lane orders, queued above: the last workgroup to finish draws them, one wavefront
    // per pixel.  Queue entries are published write-through and the count is an agent-scope atomic; every wave
    // drains its stores before the workgroup takes its arrival ticket (counters[6], zero between launches).
    K2_STAMP(3)
    if (BUILD && ride.on) { k3_ride rd = ride; if (sc.win_key) rd.d_pose = s_wpose; k2_ride_tail(rd, ride_cells, ride_ray, ride_cell, ride_r, ride_nw, ride_h, ride_nh, ride_v, ride_p); }
    if (sc.span) k2_row_spans<T>(byidx, n_rays, x1, y1, size, n_pix_wgs, sc.span);    // (only while a host mirror is being kept: slamhip_cs_holemap_mirror_async)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    K2_STAMP(4)
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(&counters[6], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == n_pix_wgs - 1;
        if (s_last) __hip_atomic_store(&counters[6], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    K2_STAMP(5)
    if (!s_last) return;
    int n_conf = __hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n_conf > cap_conflict) n_conf = cap_conflict;
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&counters[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (at rest between launches, like the ticket)
    if (threadIdx.x == 0) counters[5] = n_conf;                     // (developer statistics: SLAMHIP_K2_STATS)
    for (int item = wv; item < n_conf; item += 16) {
        const int ptr = __hip_atomic_load(&conflict_pix[item], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        k2_wave_pixel<T>(ptr % size, ptr / size, x1, y1, size, byidx, vps, n_rays, cand, start, map, alpha, sval[wv]);
    }
}

// ---- host side ----------------------------------------------------------------------------------------
int32_t cs_holemap_alloc(slamhip_cs *cs)
{
    const size_t npix = (size_t)cs->hs * cs->hs;
    SH_HIP(hipMalloc(&cs->d_k2_counters, sizeof(int) * 8));
    SH_HIP(hipMemsetAsync(cs->d_k2_counters, 0, sizeof(int) * 8, cs->ctx->stream));
    SH_HIP(hipMalloc(&cs->d_k2_start, sizeof(int) * (4 * K2_NBUCK + 1)));
    cs->cap_conflict = (int)(npix < (1u << 22) ? npix : (1u << 22));
    SH_HIP(hipMalloc(&cs->d_conflict_pix, sizeof(int) * (size_t)cs->cap_conflict));
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
    (void)hipFree(cs->d_k2_counters); (void)hipFree(cs->d_conflict_pix); (void)hipFree(cs->d_hole_dirty);
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
        SH_HIP(hipMalloc(&cs->d_rays, sizeof(k2_byidx) * (size_t)cap));
        SH_HIP(hipMalloc(&cs->d_k2_cand, sizeof(k2_cand) * (size_t)cap));
        SH_HIP(hipMalloc(&cs->d_k2_vprof, sizeof(k2_vprof) * (size_t)cap));
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
    static const int rb_env = getenv("SLAMHIP_K2_RB") ? atoi(getenv("SLAMHIP_K2_RB")) : 12, rc_env = getenv("SLAMHIP_K2_RC") ? atoi(getenv("SLAMHIP_K2_RC")) : 28;
    sc.rb_num = rb_env; sc.rc_num = rc_env;                        // (radii, per 1080 rays, from which a wavefront takes two / four zone pixels)
    {
        sh_timer t(ctx, SLAMHIP_K_CS_HOLEMAP);
        if (!build)
            hipLaunchKernelGGL(k2_prepare, dim3(1), dim3(1024), 0, ctx->stream, cs->d_pts, n, cs->hs, cs->hscale, d_pose, h_pxcs,
                               hole_width, (k2_byidx *)cs->d_rays, (k2_cand *)cs->d_k2_cand, (k2_vprof *)cs->d_k2_vprof, cs->d_k2_start, cs->d_k2_counters, (int *)cs->d_key + 6, cs->d_hole_dirty);
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
#define K2_PIXELS(B, T) {                                                                                                   \
            static std::atomic<unsigned long long> attr_set{0};              /* one bit per device (the attribute is the device's) */   \
            if (!((attr_set.load(std::memory_order_acquire) >> (ctx->device & 63)) & 1ull)) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k2_pixels<B, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)k2_lds_bytes(B, B ? K2_LDS_RAYS : 0)); attr_set.fetch_or(1ull << (ctx->device & 63), std::memory_order_release); } \
            hipLaunchKernelGGL((k2_pixels<B, T>), dim3(grid), dim3(1024), k2_lds_bytes(B, n), ctx->stream, sc, (const k2_byidx *)cs->d_rays, \
                               (const k2_vprof *)cs->d_k2_vprof, (const k2_cand *)cs->d_k2_cand, n, (const int *)cs->d_k2_start, cs->d_k2_counters, \
                               cs->hs, cs->d_hole, quality, cs->d_conflict_pix, cs->cap_conflict, grid, ride); }
        if (build) { if (cs->hs <= 16384) K2_PIXELS(true, int) else K2_PIXELS(true, long long) }
        else       { if (cs->hs <= 16384) K2_PIXELS(false, int) else K2_PIXELS(false, long long) }
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
