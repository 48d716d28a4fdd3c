// holemap.hip -- K2: HoleMap raster update, bit-exact and ray-order exact (gfx950 only).
//
// Replaces UpdateHoleMap (CoreSLAM/CoreSLAMProcessor.cs:496-534), DrawLaserRayOnHoleMap (:359-443) and
// ClipRay (:320-345).  The reference draws ray after ray with a read-modify-write blend
//     pix = (ushort)(((256 - alpha) * pix + alpha * pixval) >> 8)                     (:431)
// which does not commute for different pixval, so pixels touched by several rays must see their
// fragments in ray order.  Design:
//   setup    one thread per ray: literal float/int arithmetic of :519-530, :361-399 (clip, major
//            axis, V-profile parameters) -> ray table; rays are cut into 64-step chunks
//   count    one lane per fragment (closed-form Bresenham position, literal V-profile recurrence
//            restarted at the edge of the hole zone): atomicAdd(cnt[pixel])
//   minmax   fragments of multi-touched pixels: atomicMin / atomicMax of pixval
//   apply    single-touched pixels: blend directly.  Multi-touched pixels: one elected fragment
//            applies the blend cnt times if every fragment carried the same pixval (order-free),
//            else queues the pixel on the conflict list
//   resolve  one wavefront per conflict pixel: test all rays in ray order (closed form), apply the
//            matching fragments' blends in that order
// Closed form: after i iterations of the error recurrence (:394-396,:433-441) the walk has taken
//   m(i) = min(i, max(0, ceil((2*dyc*i - dxc) / (2*dxc))))   minor steps     (tests/test_closed_forms.py)
// All integer arithmetic wraps like C# unchecked int; float->int follows cvttss2si (sh_f2i).
// Deviations from the reference (all in exception / platform-dependent territory; the oracle does the same):
//   D1 non-representable pixel coordinates (NaN/inf, e.g. zero-range point) skip the ray;
//   D2 Math.Abs(int.MinValue) / int.MinValue / -1 skip the ray;
//   D4 a clipped endpoint outside the map (reachable only through int32 overflow in :329/:340) skips the ray.
#include "cs_internal.h"
#include "det_trig.h"

#define TS_NO_OBSTACLE 65500
#define TS_OBSTACLE 0
#define K2_CHUNK 64

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
    int chunk0, nchunks;
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

__global__ void __launch_bounds__(256)
k2_setup(const float2 *__restrict__ pts, int n_points, int size, float scale, const float *d_pose, float4 h_pxcs,
         float hole_width, cs_ray *__restrict__ rays)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_points) return;
    cs_ray r;
    memset(&r, 0, sizeof(r));
    const float4 q = k2_pxcs(d_pose, h_pxcs, scale);
    const float px = q.x, py = q.y, c = q.z, s = q.w;
    const int x1 = sh_f2i(px), y1 = sh_f2i(py);                            // :505-506
    bool ok = !(x1 < 0 || x1 >= size || y1 < 0 || y1 >= size);             // :509-512 robot out of map
    const float2 p = pts[i];
    float x2p = c * p.x - s * p.y;                                         // :519
    float y2p = s * p.x + c * p.y;                                         // :520
    const int xp = sh_f2i(px + x2p);                                       // :521
    const int yp = sh_f2i(py + y2p);                                       // :522
    const float dist = __fsqrt_rn(x2p * x2p + y2p * y2p);                  // :524
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
            r.nchunks = (dxc + 1 + K2_CHUNK - 1) / K2_CHUNK;               // steps x = 0..dxc (:404)
        }
    }
    rays[i] = r;
}

// exclusive prefix of nchunks over the rays (single workgroup; R is a few thousand)
__global__ void __launch_bounds__(1024)
k2_scan_chunks(cs_ray *__restrict__ rays, int n, int *__restrict__ counters)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    __shared__ int total_px;
    if (threadIdx.x == 0) { carry = 0; total_px = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        int v = (i < n && rays[i].valid) ? rays[i].nchunks : 0;
        // every step x = 0..dxc of a valid ray blends exactly one pixel (:404,:431): the blended-pixel statistic is
        // a plain sum (a per-wave atomic on one address would serialise at ~12 ns each and dominate the update)
        int px = (i < n && rays[i].valid) ? rays[i].dxc + 1 : 0;
        for (int off = 32; off > 0; off >>= 1) px += __shfl_down(px, off, 64);
        if (lane == 0 && px) atomicAdd(&total_px, px);
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; w++) woff += wsum[w];
        const int excl = carry + woff + incl - v;
        if (i < n) rays[i].chunk0 = excl;
        __syncthreads();
        if (threadIdx.x == 1023) carry = excl + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) { counters[0] = carry; counters[1] = 0; counters[2] = total_px; }
}

__global__ void __launch_bounds__(256)
k2_fill_chunks(const cs_ray *__restrict__ rays, int n, int *__restrict__ chunk_ray, int *__restrict__ chunk_x0, int cap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !rays[i].valid) return;
    const int c0 = rays[i].chunk0, nc = rays[i].nchunks;
    for (int k = 0; k < nc; k++)
        if (c0 + k < cap) { chunk_ray[c0 + k] = i; chunk_x0[c0 + k] = k * K2_CHUNK; }
}

// minor steps taken before step x (closed form of :394-396,:433-441)
__device__ static inline int k2_minor(const cs_ray &r, int x)
{
    if (r.dxc <= 0) return 0;
    long long num = 2ll * r.dyc * x - r.dxc;
    if (num <= 0) return 0;
    long long den = 2ll * r.dxc;
    long long m = (num + den - 1) / den;
    return m > x ? x : (int)m;
}

// pixval at step x: literal recurrence of :406-428 restarted where it starts to change
__device__ static inline int k2_pixval(const cs_ray &r, int x)
{
    int pixval = TS_NO_OBSTACLE, errorv = r.derrorv / 2;                   // :402,:397
    if (x <= r.lim2) return pixval;                                        // :406
    int xs = r.lim2 < 0 ? 0 : r.lim2 + 1;
    for (int xi = xs; xi <= x; xi++) {
        if (xi <= r.lim1) {                                                // :408
            pixval = sh_wadd(pixval, r.incv);
            errorv = sh_wadd(errorv, r.incerrorv);
            if (errorv > r.derrorv) { pixval = sh_wadd(pixval, r.sincv); errorv = sh_wsub(errorv, r.derrorv); }
        } else {
            pixval = sh_wsub(pixval, r.incv);
            errorv = sh_wsub(errorv, r.incerrorv);
            if (errorv < 0) { pixval = sh_wsub(pixval, r.sincv); errorv = sh_wadd(errorv, r.derrorv); }
        }
    }
    return pixval;
}

__device__ static inline uint16_t k2_blend(uint16_t pix, int pixval, int alpha)
{
    return (uint16_t)(sh_wadd(sh_wmul(256 - alpha, (int)pix), sh_wmul(alpha, pixval)) >> 8);   // :431
}

// One lane per fragment.  PASS 0: count, 1: min/max for multi-touched pixels, 2: apply.
template <int PASS>
__global__ void __launch_bounds__(256)
k2_fragments(const cs_ray *__restrict__ rays, const int *__restrict__ chunk_ray, const int *__restrict__ chunk_x0,
             int *__restrict__ counters, int npix, uint16_t *__restrict__ map, uint32_t *__restrict__ cnt,
             int32_t *__restrict__ vmin, int32_t *__restrict__ vmax, int alpha,
             int *__restrict__ conflict_pix, int cap_conflict)
{
    const int chunk = blockIdx.x * (256 / K2_CHUNK) + (threadIdx.x >> 6);
    if (chunk >= counters[0]) return;                 // wave-uniform
    const cs_ray r = rays[chunk_ray[chunk]];
    const int x = chunk_x0[chunk] + (threadIdx.x & 63);
    int ptr = -1;
    if (x <= r.dxc) {
        ptr = sh_wadd(sh_wadd(r.ptr0, sh_wmul(x, r.incmaj)), sh_wmul(k2_minor(r, x), r.incmin));
        if (ptr < 0 || ptr >= npix) ptr = -1;         // cannot happen for a clipped ray; guards the array
    }
    if (PASS == 0) {
        if (ptr >= 0) atomicAdd(&cnt[ptr], 1u);
        return;
    }
    const uint32_t n = ptr >= 0 ? cnt[ptr] : 0u;
    if (PASS == 1) {
        if (n > 1) {
            const int v = k2_pixval(r, x);
            atomicMin(&vmin[ptr], v);
            atomicMax(&vmax[ptr], v);
        }
        return;
    }
    // PASS 2
    if (n == 1) {
        map[ptr] = k2_blend(map[ptr], k2_pixval(r, x), alpha);
        cnt[ptr] = 0;
    } else if (n > 1) {
        const uint32_t won = atomicExch(&cnt[ptr], 0u);      // elect one fragment per pixel
        if (won != 0) {
            const int lo = vmin[ptr], hi = vmax[ptr];
            vmin[ptr] = INT32_MAX; vmax[ptr] = INT32_MIN;
            if (lo == hi) {                                  // same pixval from every ray: order-free
                uint16_t pix = map[ptr];
                for (uint32_t k = 0; k < won; k++) pix = k2_blend(pix, lo, alpha);
                map[ptr] = pix;
            } else {
                const int slot = atomicAdd(&counters[1], 1);
                if (slot < cap_conflict) conflict_pix[slot] = ptr;
            }
        }
    }
}

// One wavefront per conflict pixel: find the rays that touch it, in ray order, and blend in that order.
__global__ void __launch_bounds__(256)
k2_resolve(const cs_ray *__restrict__ rays, int n_rays, const int *__restrict__ counters, int size,
           uint16_t *__restrict__ map, int alpha, const int *__restrict__ conflict_pix, int cap_conflict)
{
    int n = counters[1];
    if (n > cap_conflict) n = cap_conflict;
    const int lane = threadIdx.x & 63;
    for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < n; w += gridDim.x * 4) {
    const int ptr = conflict_pix[w];
    const int X = ptr % size, Y = ptr / size;
    uint16_t pix = map[ptr];
    for (int base = 0; base < n_rays; base += 64) {
        const int i = base + lane;
        bool hit = false;
        int v = 0;
        if (i < n_rays) {
            const cs_ray r = rays[i];
            if (r.valid) {
                const int a = r.major_x ? X - r.x1 : Y - r.y1;
                const int b = r.major_x ? Y - r.y1 : X - r.x1;
                int x = -1;
                if (r.smaj != 0) x = a * r.smaj; else if (a == 0) x = 0;
                if (x >= 0 && x <= r.dxc) {
                    const int m = k2_minor(r, x);
                    if (m * r.smin == b) { hit = true; v = k2_pixval(r, x); }
                }
            }
        }
        unsigned long long mask = __ballot(hit);
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            const int vv = __shfl(v, src, 64);
            pix = k2_blend(pix, vv, alpha);
            mask &= mask - 1;
        }
    }
    if (lane == 0) map[ptr] = pix;
    }
}

// ---- host side ----------------------------------------------------------------------------------------
int32_t cs_holemap_alloc(slamhip_cs *cs)
{
    const size_t npix = (size_t)cs->hs * cs->hs;
    SH_HIP(hipMalloc(&cs->d_h_cnt, sizeof(uint32_t) * npix));
    SH_HIP(hipMalloc(&cs->d_h_vmin, sizeof(int32_t) * npix));
    SH_HIP(hipMalloc(&cs->d_h_vmax, sizeof(int32_t) * npix));
    SH_HIP(hipMemsetAsync(cs->d_h_cnt, 0, sizeof(uint32_t) * npix, cs->ctx->stream));
    // INT32_MAX = 0x7FFFFFFF / INT32_MIN = 0x80000000 are not byte patterns: fill with a kernel-free trick
    std::vector<int32_t> tmp(npix, INT32_MAX);
    SH_HIP(hipMemcpy(cs->d_h_vmin, tmp.data(), sizeof(int32_t) * npix, hipMemcpyHostToDevice));
    std::fill(tmp.begin(), tmp.end(), INT32_MIN);
    SH_HIP(hipMemcpy(cs->d_h_vmax, tmp.data(), sizeof(int32_t) * npix, hipMemcpyHostToDevice));
    SH_HIP(hipMalloc(&cs->d_k2_counters, sizeof(int) * 4));
    SH_HIP(hipMemsetAsync(cs->d_k2_counters, 0, sizeof(int) * 4, cs->ctx->stream));
    cs->cap_conflict = (int)npix;
    SH_HIP(hipMalloc(&cs->d_conflict_pix, sizeof(int) * (size_t)cs->cap_conflict));
    return SLAMHIP_OK;
}

void cs_holemap_free(slamhip_cs *cs)
{
    (void)hipFree(cs->d_h_cnt); (void)hipFree(cs->d_h_vmin); (void)hipFree(cs->d_h_vmax);
    (void)hipFree(cs->d_rays); (void)hipFree(cs->d_chunk_ray); (void)hipFree(cs->d_chunk_x0);
    (void)hipFree(cs->d_k2_counters); (void)hipFree(cs->d_conflict_pix);
}

int32_t cs_launch_holemap_update(slamhip_cs *cs, const float *d_pose, float4 h_pxcs, float hole_width, int quality)
{
    slamhip_ctx *ctx = cs->ctx;
    const int n = cs->n_points;
    if (n <= 0) return SLAMHIP_OK;
    if (n > cs->cap_rays) {
        if (cs->d_rays) (void)hipFree(cs->d_rays);
        cs->d_rays = nullptr; cs->cap_rays = 0;
        SH_HIP(hipMalloc(&cs->d_rays, sizeof(cs_ray) * (size_t)(n + n / 4 + 64)));
        cs->cap_rays = n + n / 4 + 64;
    }
    const int max_chunks_per_ray = (cs->hs + K2_CHUNK - 1) / K2_CHUNK;     // dxc + 1 <= Size
    const long long want = (long long)n * max_chunks_per_ray;
    if (want > cs->cap_chunks) {
        if (cs->d_chunk_ray) (void)hipFree(cs->d_chunk_ray);
        if (cs->d_chunk_x0) (void)hipFree(cs->d_chunk_x0);
        cs->d_chunk_ray = cs->d_chunk_x0 = nullptr; cs->cap_chunks = 0;
        SH_HIP(hipMalloc(&cs->d_chunk_ray, sizeof(int) * (size_t)want));
        SH_HIP(hipMalloc(&cs->d_chunk_x0, sizeof(int) * (size_t)want));
        cs->cap_chunks = (int)want;
    }
    const int npix = cs->hs * cs->hs;
    sh_timer t(ctx, SLAMHIP_K_CS_HOLEMAP);
    hipLaunchKernelGGL(k2_setup, dim3(sh_div_up(n, 256)), dim3(256), 0, ctx->stream,
                       cs->d_pts, n, cs->hs, cs->hscale, d_pose, h_pxcs, hole_width, cs->d_rays);
    hipLaunchKernelGGL(k2_scan_chunks, dim3(1), dim3(1024), 0, ctx->stream, cs->d_rays, n, cs->d_k2_counters);
    hipLaunchKernelGGL(k2_fill_chunks, dim3(sh_div_up(n, 256)), dim3(256), 0, ctx->stream,
                       cs->d_rays, n, cs->d_chunk_ray, cs->d_chunk_x0, cs->cap_chunks);
    const dim3 fgrid(sh_div_up((int)want, 256 / K2_CHUNK));
    hipLaunchKernelGGL(k2_fragments<0>, fgrid, dim3(256), 0, ctx->stream, cs->d_rays, cs->d_chunk_ray, cs->d_chunk_x0,
                       cs->d_k2_counters, npix, cs->d_hole, cs->d_h_cnt, cs->d_h_vmin, cs->d_h_vmax, quality,
                       cs->d_conflict_pix, cs->cap_conflict);
    hipLaunchKernelGGL(k2_fragments<1>, fgrid, dim3(256), 0, ctx->stream, cs->d_rays, cs->d_chunk_ray, cs->d_chunk_x0,
                       cs->d_k2_counters, npix, cs->d_hole, cs->d_h_cnt, cs->d_h_vmin, cs->d_h_vmax, quality,
                       cs->d_conflict_pix, cs->cap_conflict);
    hipLaunchKernelGGL(k2_fragments<2>, fgrid, dim3(256), 0, ctx->stream, cs->d_rays, cs->d_chunk_ray, cs->d_chunk_x0,
                       cs->d_k2_counters, npix, cs->d_hole, cs->d_h_cnt, cs->d_h_vmin, cs->d_h_vmax, quality,
                       cs->d_conflict_pix, cs->cap_conflict);
    // conflict pixels are rare (0.015 % of touched pixels at 2048^2); the grid is sized for
    // the worst case and exits on the device-side count
    const int rgrid = sh_div_up(cs->cap_conflict < 16384 ? cs->cap_conflict : 16384, 4);
    hipLaunchKernelGGL(k2_resolve, dim3(rgrid), dim3(256), 0, ctx->stream, cs->d_rays, n, cs->d_k2_counters, cs->hs,
                       cs->d_hole, quality, cs->d_conflict_pix, cs->cap_conflict);
    SH_HIP(hipGetLastError());
    return SLAMHIP_OK;
}
