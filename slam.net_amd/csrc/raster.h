// raster.h -- shared pieces of the pixel-centric line rasterisers (K2 HoleMap update, K5 Hector grid update).
//
// A Bresenham-type walk from a common start cell takes, at its i-th step, i cells along its major axis and
// m(i) cells along the minor one with |m(i) - slope * i| <= 1/2 (slope = minor length / major length): a cell at
// major offset a and signed minor offset b can only be drawn by lines of its direction class (major axis and its
// sign) whose signed slope lies in [(b - 1/2) / a, (b + 1/2) / a].  Lines are counting-sorted by (class, slope bucket),
// so the candidates of a cell are one contiguous range of that table; the exact closed-form test follows.
#pragma once
#include <hip/hip_runtime.h>

#define RS_NBUCK 1024                  // slope buckets per direction class

__device__ static inline int rs_bucket(float t)
{
    int k = (int)floorf((t + 1.0f) * (RS_NBUCK / 2));
    return k < 0 ? 0 : k > RS_NBUCK - 1 ? RS_NBUCK - 1 : k;
}
// direction class of a line: 0 E, 1 W (x major), 2 S, 3 N (y major)
__device__ static inline int rs_class(bool major_x, int smaj) { return major_x ? (smaj >= 0 ? 0 : 1) : (smaj >= 0 ? 2 : 3); }

// candidate range of cell (a, b) in one class.  The walks take their m-th minor step where slope * i passes m - 1/2
// (K2: m(i) = ceil(slope * i - 1/2), CoreSLAMProcessor.cs:394-441; K5: m(i) = floor((da / 2 + i * db) / da),
// OccGridMap.cs:220-239, whose integer da / 2 adds up to 1 / (2 da) <= 1 / (2 a) to slope * i -- `extra`, in slope units), so a
// line that draws (a, b) has its signed slope in [(b - 1/2) / a - extra, (b + 1/2) / a + extra].  The margin covers the
// float roundings of this range and of the slopes the lines were bucketed with (each below 2e-7; a bucket is 2e-3 wide).
// (1 / a is the hardware reciprocal, within 1 ulp: a slope error below 1e-7, inside the margin; `extra_a2` is the extra term
// times a^2 -- 0 for K2, 1/2 for K5 -- so that the callers need no division of their own)
template <typename ST>       // (the bucket table: int, or unsigned short where LDS is tight)
__device__ static inline void rs_range(const ST *start, int cls, int a, int b, float extra_a2, int &lo, int &hi)
{
    const float ra = __builtin_amdgcn_rcpf((float)a), m = extra_a2 * ra * ra + 4.0e-6f;
    const int blo = rs_bucket(((float)b - 0.5f) * ra - m), bhi = rs_bucket(((float)b + 0.5f) * ra + m);
    lo = (int)start[cls * RS_NBUCK + blo];
    hi = (int)start[cls * RS_NBUCK + bhi + 1];
}
// classes of a cell at offset (dx, dy) from the start cell; a diagonal cell has two; the start cell itself none
__device__ static inline int rs_classes(int dx, int dy, int cls[2], int a[2], int b[2])
{
    // (no run-time subscripts: with `cls[n++] = ...` the compiler kept the three arrays in scratch memory -- a memory round
    // trip at the head of every pixel's lookup)
    const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
    const bool cx = adx >= ady && adx > 0, cy = ady >= adx && ady > 0;
    const int clsx = dx > 0 ? 0 : 1, clsy = dy > 0 ? 2 : 3;
    cls[0] = cx ? clsx : clsy; a[0] = cx ? adx : ady; b[0] = cx ? dy : dx;
    cls[1] = clsy;             a[1] = ady;            b[1] = dx;              // (the second class of a diagonal cell)
    return (cx ? 1 : 0) + (cy ? 1 : 0);
}
