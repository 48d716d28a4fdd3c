// holemap.hip -- K2: HoleMap raster update, bit-exact and ray-order exact (gfx950 only).
//
// Replaces UpdateHoleMap (CoreSLAM/CoreSLAMProcessor.cs:496-534), DrawLaserRayOnHoleMap (:359-443) and
// ClipRay (:320-345).  The reference draws ray after ray with a read-modify-write blend
//     pix = (ushort)(((256 - alpha) * pix + alpha * pixval) >> 8)                     (:431)
// which does not commute for different pixval, so a pixel touched by several rays must see their
// fragments in ray order.  The update is PIXEL-centric (a gather, no atomics on the map,
// no per-pixel scratch): a pixel asks which rays draw it, then blends their values in ray order.
//   prepare  one workgroup: per ray the literal float/int arithmetic of :519-530, :361-399 (clip, major axis,
//            V-profile parameters) -> ray table; rays are counting-sorted into 4 direction classes (major axis
//            and its sign) x 1024 buckets of signed slope (minor / major)
//   pixels   one lane per pixel of the scan's bounding square: step x of a ray lies at major offset x and minor
//            offset m(x) = min(x, max(0, ceil((2*dyc*x - dxc) / (2*dxc)))) (closed form of the error
//            recurrence :394-396,:433-441; tests/test_closed_forms.py), and |m(x) - slope*x| <= 1/2, so only
//            rays of the pixel's class with slope in [(b-1)/a, (b+1)/a] can draw pixel (major a, minor b): a
//            contiguous range of the sorted table, tested exactly.  Up to 4 hits are sorted by ray index in
//            registers and blended; a pixel with more goes to the conflict list
//            (same launch) one wavefront per pixel for the neighbourhood of the robot (every ray passes there) and,
//            in the last workgroup to finish, for the conflict list: lanes test the candidate rays, hits are
//            rank-sorted by ray index and blended in that order; the few pixels with more than 64 candidates scan
//            all rays in index order
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
__device__ static inline int k2_pixval_closed(const k2_vprof p, int x)
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

__device__ static inline uint16_t k2_blend(uint16_t pix, int pixval, int alpha)
{
    return (uint16_t)(sh_wadd(sh_wmul(256 - alpha, (int)pix), sh_wmul(alpha, pixval)) >> 8);   // :431
}

// does step x = a (major offset a >= 1) of the ray draw the pixel at signed minor offset b?  (closed form, no division;
// T = int for maps up to 16384 pixels a side -- 2*dyc*a < 2^29 -- else long long: 64-bit multiplies are several
// quarter-rate instructions each)
template <typename T>
__device__ static inline bool k2_hit(const k2_cand c, int a, int b)
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
           k2_byidx *__restrict__ byidx, k2_cand *__restrict__ cand, k2_vprof *__restrict__ vprof, int *__restrict__ start,
           int *__restrict__ counters, int *__restrict__ total_out, int *__restrict__ dirty, const k3_ride ride)
{
    if (blockIdx.x > 0) {                                          // riding along: the ray walks of the ObstacleMap update
        k3_rays_unit((blockIdx.x - 1) * 16 + (threadIdx.x >> 6), threadIdx.x & 63, ride.pts, ride.n_points, ride.size, ride.scale,
                     ride.d_pose, ride.h_pxcs, ride.hits, ride.nohit, ride.chunks_per_ray);
        return;
    }
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
template <typename CT>
__device__ static inline void k2_wave_pixel(int X, int Y, int x1, int y1, int size, const k2_byidx *__restrict__ byidx,
                                            const k2_vprof *__restrict__ vprof, int n_rays, CT cand, const int *start,
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
                    if (x == 0 ? bb == 0 : (x > 0 && k2_hit<long long>(c, x, bb))) {
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
            if (k2_hit<long long>(c, aa, bb)) { hit = true; idx = c.ray; v = aa <= c.lim2 ? TS_NO_OBSTACLE : k2_pixval_closed(vprof[c.ray], aa); }
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

// lane per pixel over the bounding square of the scan (persistent grid; a wavefront takes 64 pixels of one row).
// The bucket table and -- when it fits (LDS_TABLE) -- the candidate table live in LDS: a pixel's lookup is a chain of
// dependent small reads (bucket bounds -> candidates -> map), which global-memory latency would dominate.
#ifndef K2_LDS_RAYS
#define K2_LDS_RAYS 3072
#endif
#ifdef K2_TIMES
// developer instrumentation (build with SLAMHIP_K2_TIMES=1): 100 MHz wall-clock stamps per workgroup and phase
__device__ unsigned long long g_k2_times[512 * 8];
#define K2_STAMP(k) { if (threadIdx.x == 0 && blockIdx.x < 512) g_k2_times[blockIdx.x * 8 + (k)] = wall_clock64(); }
#else
#define K2_STAMP(k) {}
#endif
template <bool LDS_TABLE, typename T>
__global__ void __launch_bounds__(1024)
k2_pixels(const k2_byidx *__restrict__ byidx, const k2_vprof *__restrict__ vprof, const k2_cand *__restrict__ cand_g, int n_rays,
          const int *__restrict__ start_g, int *__restrict__ counters, int size, uint16_t *__restrict__ map, int alpha,
          int *__restrict__ conflict_pix, int cap_conflict, int n_pix_wgs, const k3_ride ride)
{
    if ((int)blockIdx.x >= n_pix_wgs) {                            // riding along: the cell pass of the ObstacleMap update
        k3_apply_cell(((int)blockIdx.x - n_pix_wgs) * 1024 + threadIdx.x, ride.map, ride.n_cells, ride.hits, ride.nohit, ride.max_hits);
        return;
    }
    __shared__ int start[4 * K2_NBUCK + 1];
    __shared__ __attribute__((aligned(16))) k2_cand cand_s[LDS_TABLE ? K2_LDS_RAYS : 1];
    __shared__ int sval[16][64];
    __shared__ int s_last, s_next_zone, s_next_item;
    K2_STAMP(0)
    const int R = counters[0], x1 = counters[3], y1 = counters[4];
    if (x1 < 0 || x1 >= size || y1 < 0 || y1 >= size) return;       // robot outside the map: nothing is drawn (:509-512)
    int X0 = max(x1 - R, 0), X1 = min(x1 + R, size - 1), Y0 = max(y1 - R, 0), Y1 = min(y1 + R, size - 1);
    if (threadIdx.x == 0) { s_next_zone = 0; s_next_item = 0; }
    for (int i = threadIdx.x; i <= 4 * K2_NBUCK; i += 1024) start[i] = start_g[i];
    if (LDS_TABLE) for (int i = threadIdx.x; i < n_rays; i += 1024) cand_s[i] = cand_g[i];     // (entries past the valid rays are never addressed)
    __syncthreads();
    K2_STAMP(1)
    const k2_cand *cand = LDS_TABLE ? cand_s : cand_g;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Work items are dealt to the workgroups round-robin (every workgroup gets pixels from all over the square) and
    // inside a workgroup to whichever wavefront is free (an LDS counter): the cost of an item depends on how many
    // rays cross its pixels, and a workgroup is only as fast as its slowest wavefront.
    // (1) the zone around the robot (Chebyshev distance <= Z): one wavefront per pixel
    const int Z = K2_ZONE - 1 < R ? K2_ZONE - 1 : R;
    const int side = 2 * Z + 1, n_zone = side * side;
    for (;;) {
        int k = 0;
        if (lane == 0) k = atomicAdd(&s_next_zone, 1);
        k = __builtin_amdgcn_readfirstlane(k);
        const int item = blockIdx.x + k * n_pix_wgs;
        if (item >= n_zone) break;
        const int X = x1 - Z + item % side, Y = y1 - Z + item / side;
        if (X < 0 || X >= size || Y < 0 || Y >= size) continue;              // wave-uniform
        k2_wave_pixel(X, Y, x1, y1, size, byidx, vprof, n_rays, cand, start, map, alpha, sval[wv]);
    }
    K2_STAMP(2)
    // (2) the rest of the bounding square: one lane per pixel, a wavefront takes 64 pixels of one row
    const int tiles_x = (X1 - X0 + 64) / 64, items = R > 0 ? tiles_x * (Y1 - Y0 + 1) : 0;
    for (;;) {
        int k = 0;
        if (lane == 0) k = atomicAdd(&s_next_item, 1);
        k = __builtin_amdgcn_readfirstlane(k);
        const int item = blockIdx.x + k * n_pix_wgs;
        if (item >= items) break;
        const int row = item / tiles_x, tx = item - row * tiles_x;
        const int X = X0 + tx * 64 + lane, Y = Y0 + row;
        if (X > X1) continue;
        const int dx = X - x1, dy = Y - y1;
        if (max(dx < 0 ? -dx : dx, dy < 0 ? -dy : dy) < K2_ZONE) continue;     // (1)'s pixels
        const int ptr = Y * size + X;
        const uint16_t pix_in = map[ptr];                          // requested now, needed after the search (no other lane touches this pixel)
        int cls[2], a[2], b[2];
        const int ncls = rs_classes(dx, dy, cls, a, b);
        int hidx[K2_MAXHIT], hval[K2_MAXHIT], nh = 0;
        bool overflow = false;
        for (int k = 0; k < ncls; k++) {
            int lo, hi;
            rs_range(start, cls[k], a[k], b[k], 0.0f, lo, hi);
            for (int ci = lo; ci < hi; ci++) {
                const k2_cand c = cand[ci];
                if (!k2_hit<T>(c, a[k], b[k])) continue;
                const int v = a[k] <= c.lim2 ? TS_NO_OBSTACLE : k2_pixval_closed(vprof[c.ray], a[k]);
                if (nh == K2_MAXHIT) { overflow = true; break; }
                // insert by ray index (the list stays sorted; compile-time subscripts keep it in registers)
                int posn = 0;
#pragma unroll
                for (int s = 0; s < K2_MAXHIT; s++) if (s < nh && hidx[s] < c.ray) posn++;
#pragma unroll
                for (int s = K2_MAXHIT - 1; s >= 1; s--) if (s > posn && s <= nh) { hidx[s] = hidx[s - 1]; hval[s] = hval[s - 1]; }
#pragma unroll
                for (int s = 0; s < K2_MAXHIT; s++) if (s == posn) { hidx[s] = c.ray; hval[s] = v; }
                nh++;
            }
            if (overflow) break;
        }
        if (overflow) {
            const int slot = __hip_atomic_fetch_add(&counters[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (slot < cap_conflict) __hip_atomic_store(&conflict_pix[slot], ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (nh > 0) {
            uint16_t pix = pix_in;
#pragma unroll
            for (int s = 0; s < K2_MAXHIT; s++) if (s < nh) pix = k2_blend(pix, hval[s], alpha);
            map[ptr] = pix;
        }
    }
    // (3) pixels with more than K2_MAXHIT hits, queued by (2): the last workgroup to finish draws them, one wavefront
    //     per pixel.  Queue entries are published write-through and the count is an agent-scope atomic; every wave
    //     drains its stores before the workgroup takes its arrival ticket (counters[6], zero between launches).
    K2_STAMP(3)
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
    for (int item = wv; item < n_conf; item += 16) {
        const int ptr = __hip_atomic_load(&conflict_pix[item], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        k2_wave_pixel(ptr % size, ptr / size, x1, y1, size, byidx, vprof, n_rays, cand, start, map, alpha, sval[wv]);
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

// ride != nullptr: the ObstacleMap update (prepared by cs_obstacle_ride) travels in the same two launches
int32_t cs_launch_holemap_update(slamhip_cs *cs, const float *d_pose, float4 h_pxcs, float hole_width, int quality, const k3_ride *ride_in)
{
    k3_ride ride;
    if (ride_in) ride = *ride_in; else memset(&ride, 0, sizeof(ride));
    const int ride_rays = ride.n_blocks ? sh_div_up(ride.n_points * ride.chunks_per_ray, 16) : 0;
    const int ride_cells = ride.n_blocks ? sh_div_up(ride.n_cells, 1024) : 0;
    slamhip_ctx *ctx = cs->ctx;
    const int n = cs->n_points;
    if (n <= 0) return SLAMHIP_OK;
    SH_TRY(cs_flush_scan(cs));
    if (n > cs->cap_rays) {
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
    sh_timer t(ctx, SLAMHIP_K_CS_HOLEMAP);
    hipLaunchKernelGGL(k2_prepare, dim3(1 + ride_rays), dim3(1024), 0, ctx->stream, cs->d_pts, n, cs->hs, cs->hscale, d_pose, h_pxcs,
                       hole_width, (k2_byidx *)cs->d_rays, (k2_cand *)cs->d_k2_cand, (k2_vprof *)cs->d_k2_vprof, cs->d_k2_start, cs->d_k2_counters, (int *)cs->d_key + 6, cs->d_hole_dirty, ride);
    // One round of resident workgroups (a second round would start when the first drains: measured 51 -> 42 us at
    // 2048^2 together with the per-workgroup work counter): what the occupancy calculator says fits, times the CUs.
    static const int grid_env = getenv("SLAMHIP_K2_GRID") ? atoi(getenv("SLAMHIP_K2_GRID")) : 0;
#define K2_PIXELS(L, T) {                                                                                                   \
        static int per_cu = 0;                                                                                              \
        if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k2_pixels<L, T>, 1024, 0) != hipSuccess || per_cu < 1)) per_cu = 1; \
        const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;                                                              \
        const int grid = grid_env > 0 ? grid_env : (L ? 1 : per_cu < 2 ? per_cu : 2) * cus;   /* (with the LDS table one workgroup per CU measured best) */ \
        static bool told = false;                                                                                           \
        if (!told && getenv("SLAMHIP_K2_STATS")) { told = true; fprintf(stderr, "[slamhip] K2 pixel kernel: %d workgroups (%d per CU x %d CUs)\n", grid, per_cu, ctx->num_cus); } \
        hipLaunchKernelGGL((k2_pixels<L, T>), dim3(grid + ride_cells), dim3(1024), 0, ctx->stream, (const k2_byidx *)cs->d_rays, \
                           (const k2_vprof *)cs->d_k2_vprof, (const k2_cand *)cs->d_k2_cand, n, (const int *)cs->d_k2_start, cs->d_k2_counters, \
                           cs->hs, cs->d_hole, quality, cs->d_conflict_pix, cs->cap_conflict, grid, ride); }
    if (n <= K2_LDS_RAYS) { if (cs->hs <= 16384) K2_PIXELS(true, int) else K2_PIXELS(true, long long) }
    else                  { if (cs->hs <= 16384) K2_PIXELS(false, int) else K2_PIXELS(false, long long) }
#undef K2_PIXELS
    SH_HIP(hipGetLastError());
#ifdef K2_TIMES
    {
        static int calls = 0;
        if (++calls == 12) {
            (void)hipStreamSynchronize(ctx->stream);
            std::vector<unsigned long long> h(512 * 8);
            (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_k2_times), sizeof(unsigned long long) * h.size());
            // (the symbol is zeroed after the dump: only the workgroups of this launch have stamps)
            unsigned long long t0 = ~0ull, t1 = 0;
            int nb = 0;
            for (int i = 0; i < 512; i++) if (h[i * 8] && h[i * 8 + 5] >= h[i * 8]) { nb++; t0 = std::min(t0, h[i * 8]); t1 = std::max(t1, h[i * 8 + 5]); }
            static const char *nm[5] = { "tables", "zone", "outer", "drain", "ticket" };
            double acc[5] = { 0 }, mx[5] = { 0 }, smax = 0;
            for (int i = 0; i < 512; i++) if (h[i * 8] && h[i * 8 + 5] >= h[i * 8]) {
                for (int k = 0; k < 5; k++) { const double d = (double)(h[i * 8 + k + 1] - h[i * 8 + k]) * 0.01; acc[k] += d; mx[k] = std::max(mx[k], d); }
                smax = std::max(smax, (double)(h[i * 8] - t0) * 0.01);
            }
            fprintf(stderr, "[k2 times] %d workgroups, span %.2f us; first thread of each workgroup, mean (max):", nb, (double)(t1 - t0) * 0.01);
            for (int k = 0; k < 5; k++) fprintf(stderr, " %s %.2f (%.2f) |", nm[k], acc[k] / std::max(nb, 1), mx[k]);
            fprintf(stderr, " last workgroup starts at %.2f us\n", smax);
        }
        if (calls == 11) { std::vector<unsigned long long> z(512 * 8, 0ull); (void)hipStreamSynchronize(ctx->stream); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k2_times), z.data(), sizeof(unsigned long long) * z.size()); }
    }
#endif
    return SLAMHIP_OK;
}
