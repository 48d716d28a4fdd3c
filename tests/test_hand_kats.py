"""Known answers derived BY HAND from the C# text of the reference -- not from either restatement in oracle/.

The reference ships no tests or fixtures (SLAM.sln:6-15) and cannot be built in this image, so the oracle is "parity
unpinned" against the C# itself.  These cases are a third, human-checkable anchor: every expected number below was
worked out with pencil-and-paper arithmetic from CoreSLAM/CoreSLAMProcessor.cs:320-443 (ClipRay,
DrawLaserRayOnHoleMap), :496-534 (UpdateHoleMap), :226-259 (CalculateDistanceSISD), :456-490 / :540-593
(DrawLaserRayOnObstacleMap, UpdateObstacleMap), HectorSLAM/Map/OccGridMap.cs:114-239 and HectorSLAM/Matcher/ScanMatcher.cs:135-249
(GetCompleteHessianDerivs, InterpMapValueWithDerivatives) and :64-125 (MatchData, EstimateTransformationLogLh); the working is written out
in the comments so that a reader can follow it against the C# without running anything.  The C oracle, the NumPy
oracle and (on the GPU box) the HIP kernels must all reproduce them.

Conventions used in the working: C# `int / int` truncates toward zero; `(int)float` truncates toward zero;
TS_NO_OBSTACLE = 65500, TS_OBSTACLE = 0 (:21-22); a fresh HoleMap holds (0 + 65500) / 2 = 32750 (:169).
"""
import math

import numpy as np
import pytest

SIZE = 16
FRESH = 32750


# ---------------------------------------------------------------------------------------------------------------------
# Ray A -- x-major, inside the map, the whole V-profile visible, alpha = 128.
#   DrawLaserRayOnHoleMap(x1=8, y1=8, x2=14, y2=8, xp=12, yp=8, value=0, alpha=128), Size 16
#   :365-366 nothing to clip.  :368-374 dx=6 dy=0 dxc=6 dyc=0 incptrx=+1 incptry=0*16=0 sincv=Sign(0-65500)=-1
#   :377 dx>dy -> derrorv=|xp-x2|=|12-14|=2
#   :394-399 error=2*0-6=-6 horiz=0 diago=-12 errorv=2/2=1 incv=-65500/2=-32750 incerrorv=-65500-2*(-32750)=0
#   :401 ptr=8*16+8=136, pixval=65500
#   x=0,1,2: x > dx-2*derrorv = 2 is false               -> pixval 65500
#   x=3: 3>2 and 3<=dx-derrorv=4 -> pixval+=incv = 32750; errorv=1 (not > 2)
#   x=4: 4<=4                   -> pixval = 0
#   x=5: else branch            -> pixval-=incv = 32750; errorv-=0 = 1 (not < 0)
#   x=6:                        -> pixval = 65500
#   error stays -6 (+= horiz = 0): no step in y; pixels ptr 136..142
#   blend :431 with alpha 128 on 32750: (128*32750 + 128*pixval) >> 8
#       65500 -> 12 576 000 >> 8 = 49125;  32750 -> 8 384 000 >> 8 = 32750;  0 -> 4 192 000 >> 8 = 16375
RAY_A = dict(args=(8, 8, 14, 8, 12, 8, 0, 128),
             want={136: 49125, 137: 49125, 138: 49125, 139: 32750, 140: 16375, 141: 32750, 142: 49125})

# Ray B -- y-major after the swap of :381-387, clipped at the top edge with a NEGATIVE truncating division, alpha = 128.
#   Draw(x1=8, y1=8, x2=15, y2=18, xp=14, yp=16, value=0, alpha=128)
#   :365 ClipRay(16, ref x2c=15, ref y2c=18, 8, 8): 0 <= 15 < 16 -> untouched
#   :366 ClipRay(16, ref y2c=18, ref x2c=15, 8, 8): 18 >= 16, 18 != 8 ->
#        x2c += (15-8)*(16-1-18)/(18-8) = 7*(-3)/10 = -21/10 = -2 (toward zero; floor would give -3) -> x2c=13, y2c=15
#   :368-374 dx=7 dy=10 dxc=|13-8|=5 dyc=|15-8|=7 incptrx=+1 incptry=+16 sincv=-1
#   :377 dx>dy false -> dx=10, (dxc,dyc)=(7,5), (incptrx,incptry)=(16,1), derrorv=|yp-y2|=|16-18|=2
#   :394-399 error=2*5-7=3 horiz=10 diago=2*(5-7)=-4 errorv=1 incv=-32750 incerrorv=0;  ptr=136
#   hole starts at x > dx-2*derrorv = 6: only x=7 (<= dx-derrorv = 8) -> pixval 32750; x=0..6 -> 65500
#   walk (blend at ptr, then error>0 ? ptr+=incptry(1), error+=diago : error+=horiz; then ptr+=incptrx(16)):
#     x=0 ptr=136 e=3  -> 137,e=-1 -> 153 | x=1 ptr=153 e=-1 -> e=9 -> 169 | x=2 ptr=169 e=9 -> 170,e=5 -> 186
#     x=3 ptr=186 e=5  -> 187,e=1  -> 203 | x=4 ptr=203 e=1  -> 204,e=-3 -> 220 | x=5 ptr=220 e=-3 -> e=7 -> 236
#     x=6 ptr=236 e=7  -> 237,e=3  -> 253 | x=7 ptr=253 = 15*16+13: the clipped end point (13,15)
RAY_B = dict(args=(8, 8, 15, 18, 14, 16, 0, 128),
             want={136: 49125, 153: 49125, 169: 49125, 186: 49125, 203: 49125, 220: 49125, 236: 49125, 253: 32750})

# Ray C -- x-major towards -x, clipped at the left edge (quotient 0 by truncation), derrorv = 3 so that incv has a
# remainder and the rising half of the V takes its carry, alpha = 64.
#   Draw(x1=9, y1=5, x2=-2, y2=2, xp=1, yp=3, value=0, alpha=64)
#   :365 ClipRay(16, ref x2c=-2, ref y2c=2, 9, 5): -2 < 0, -2 != 9 -> y2c += (2-5)*(-(-2))/(-2-9) = (-6)/(-11) = 0 -> y2c=2, x2c=0
#   :366 y2c=2 in range.   dx=11 dy=3 dxc=9 dyc=3 incptrx=-1 incptry=-16 sincv=-1;  dx>dy -> derrorv=|1-(-2)|=3
#   error=2*3-9=-3 horiz=6 diago=2*(3-9)=-12 errorv=3/2=1 incv=-65500/3=-21833 incerrorv=-65500-3*(-21833)=-1;  ptr=5*16+9=89
#   thresholds: dx-2*derrorv=5, dx-derrorv=8
#     x=0 ptr=89 65500 e=-3->3 ->88 | x=1 ptr=88 65500 e=3 -> 72,e=-9 ->71 | x=2 ptr=71 65500 e=-9->-3 ->70
#     x=3 ptr=70 65500 e=-3->3 ->69 | x=4 ptr=69 65500 e=3 -> 53,e=-9 ->52 | x=5 ptr=52 65500 (5>5 false) e=-9->-3 ->51
#     x=6 ptr=51 pixval=65500-21833=43667 errorv=1-1=0 (not > 3)          e=-3->3 ->50
#     x=7 ptr=50 pixval=21834 errorv=-1                                    e=3 -> 34,e=-9 ->33
#     x=8 ptr=33 pixval=1 errorv=-2                                        e=-9->-3 ->32
#     x=9 ptr=32 (9>8: rising half) pixval=1+21833=21834, errorv=-2-(-1)=-1 < 0 -> pixval-=sincv = 21835, errorv=2
#   blend alpha 64 on 32750: (192*32750 + 64*pixval) >> 8, 192*32750 = 6 288 000
#     65500: 10 480 000/256 = 40937.5 -> 40937 | 43667: 9 082 688/256 = 35479.25 -> 35479 | 21834: 7 685 376/256 = 30021
#     1: 6 288 064/256 = 24562.75 -> 24562     | 21835: 7 685 440/256 = 30021.25 -> 30021
RAY_C = dict(args=(9, 5, -2, 2, 1, 3, 0, 64),
             want={89: 40937, 88: 40937, 71: 40937, 70: 40937, 69: 40937, 52: 40937, 51: 35479, 50: 30021, 33: 24562, 32: 30021})

# Ray D -- clipped by BOTH ClipRay calls (:365 on x with the quotient of y, then :366 with the axes swapped on the already clipped
# end point), both quotients negative and truncated toward zero (floor would give one less each time); y-major, alpha = 128.
#   Draw(x1=8, y1=8, x2=20, y2=22, xp=17, yp=18, value=0, alpha=128)
#   :365 ClipRay(16, ref x2c=20, ref y2c=22, 8, 8): 20 >= 16, 20 != 8 ->
#        y2c += (22-8)*(16-1-20)/(20-8) = 14*(-5)/12 = -70/12 = -5 (floor: -6) -> y2c=17, x2c=15
#   :366 ClipRay(16, ref y2c=17, ref x2c=15, 8, 8): 17 >= 16, 17 != 8 ->
#        x2c += (15-8)*(16-1-17)/(17-8) = 7*(-2)/9 = -14/9 = -1 (floor: -2) -> x2c=14, y2c=15:  the drawn ray ends at (14, 15)
#   :368-374 dx=12 dy=14 dxc=|14-8|=6 dyc=|15-8|=7 incptrx=+1 incptry=+16 sincv=-1
#   :377 dx>dy false -> dx=14, (dxc,dyc)=(7,6), (incptrx,incptry)=(16,1), derrorv=|yp-y2|=|18-22|=4
#   :394-399 error=2*6-7=5 horiz=12 diago=2*(6-7)=-2 errorv=4/2=2 incv=-65500/4=-16375 incerrorv=-65500-4*(-16375)=0;  ptr=136
#   hole: x > dx-2*derrorv = 6 and x <= dx-derrorv = 10: only x=7 is drawn of it -> pixval 65500-16375 = 49125; x=0..6 -> 65500
#   walk (blend at ptr, then error>0 ? ptr+=1, error-=2 : error+=12; then ptr+=16):
#     x=0 ptr=136 e=5 -> 137,e=3 -> 153 | x=1 ptr=153 e=3 -> 154,e=1 -> 170 | x=2 ptr=170 e=1 -> 171,e=-1 -> 187
#     x=3 ptr=187 e=-1 -> e=11 -> 203   | x=4 ptr=203 e=11 -> 204,e=9 -> 220 | x=5 ptr=220 e=9 -> 221,e=7 -> 237
#     x=6 ptr=237 e=7 -> 238,e=5 -> 254 | x=7 ptr=254 = 15*16+14: the twice-clipped end point (14,15)
#   blend alpha 128 on 32750: 65500 -> 49125;  49125 -> (128*32750 + 128*49125) >> 8 = 10 480 000 >> 8 = 40937 (.5 dropped)
RAY_D = dict(args=(8, 8, 20, 22, 17, 18, 0, 128),
             want={136: 49125, 153: 49125, 170: 49125, 187: 49125, 203: 49125, 220: 49125, 237: 49125, 254: 40937})

# Rays A then B on ONE map: pixel 136 (the robot's) is drawn twice, in ray order (:431 does not commute):
#   after A: 49125; B blends 65500 onto it: (128*49125 + 128*65500) >> 8 = 14 672 000 >> 8 = 57312 (.5 dropped)
A_THEN_B_136 = 57312


def _expect(want):
    m = np.full(SIZE * SIZE, FRESH, np.uint16)
    for k, v in want.items():
        m[k] = v
    return m


@pytest.mark.parametrize("ray", [RAY_A, RAY_B, RAY_C, RAY_D], ids=["A", "B", "C", "D"])
def test_hand_rays_c_oracle(oc, ray):
    pix = np.full(SIZE * SIZE, FRESH, np.uint16)
    oc.draw_ray_holemap(pix, SIZE, *ray["args"])
    assert (pix == _expect(ray["want"])).all(), np.flatnonzero(pix != _expect(ray["want"]))


@pytest.mark.parametrize("ray", [RAY_A, RAY_B, RAY_C, RAY_D], ids=["A", "B", "C", "D"])
def test_hand_rays_numpy_oracle(npo, ray):
    x1, y1, x2, y2, xp, yp, value, alpha = ray["args"]
    frags = npo.ray_fragments(SIZE, x1, y1, x2, y2, xp, yp, value)
    pix = np.full(SIZE * SIZE, FRESH, np.uint16)
    for ptr, pixval in frags:
        pix[ptr] = npo.blend(int(pix[ptr]), pixval, alpha)
    assert (pix == _expect(ray["want"])).all()


def test_hand_clip_quotients(oc):
    # the two truncating divisions worked out above
    assert oc.clip_ray(16, 18, 15, 8, 8) == (True, 15, 13)            # ray B, second ClipRay: x2c 15 -> 13, y2c 18 -> 15
    assert oc.clip_ray(16, -2, 2, 9, 5) == (True, 0, 2)               # ray C, first ClipRay: (-6)/(-11) = 0
    assert oc.clip_ray(16, 20, 22, 8, 8) == (True, 15, 17)            # ray D, first ClipRay: -70/12 = -5 (not -6)
    assert oc.clip_ray(16, 17, 15, 8, 8) == (True, 15, 14)            # ray D, second ClipRay on the clipped point, axes swapped: -14/9 = -1 (not -2)


def test_hand_ray_order_c_oracle(oc):
    pix = np.full(SIZE * SIZE, FRESH, np.uint16)
    oc.draw_ray_holemap(pix, SIZE, *RAY_A["args"])
    oc.draw_ray_holemap(pix, SIZE, *RAY_B["args"])
    want = _expect({**RAY_A["want"], **RAY_B["want"]})
    want[136] = A_THEN_B_136
    assert (pix == want).all()


# ---------------------------------------------------------------------------------------------------------------------
# The same two rays through UpdateHoleMap (:496-534), so that the float part is covered too -- with numbers that are exact
# in binary32, or whose truncation is far from an integer:
#   map 16 px over 8 m -> Scale 2;  Pose (4, 4, 0) -> px = py = 4*2 + 0.5 = 8.5, c = cos(0)*2 = 2, s = 0;  x1 = y1 = 8
#   HoleWidth 2.0 m, Quality 128
#   point (2, 0):  x2p = 2*2 - 0*0 = 4, y2p = 0; xp = (int)12.5 = 12, yp = (int)8.5 = 8; dist = 4;
#                  add = 2.0*2/2/4 = 0.5 -> x2p = 6 -> x2 = (int)14.5 = 14, y2 = 8                       == ray A
#   point (3, 4):  x2p = 6, y2p = 8; xp = (int)14.5 = 14, yp = (int)16.5 = 16; dist = sqrt(36 + 64) = 10;
#                  add = 2/10 = 0.2 -> x2p = 6*1.2 = 7.2(0000005), y2p = 9.6(000001) -> x2 = (int)15.7 = 15, y2 = (int)18.1 = 18   == ray B
UPDATE_XY = np.array([[2.0, 0.0], [3.0, 4.0]], np.float32)
UPDATE_POSE = np.array([4.0, 4.0, 0.0], np.float32)


def _update_want():
    want = _expect({**RAY_A["want"], **RAY_B["want"]})
    want[136] = A_THEN_B_136
    return want


def test_hand_update_c_oracle(oc):
    pix = np.full(SIZE * SIZE, FRESH, np.uint16)
    n = oc.update_holemap(pix, SIZE, 2.0, UPDATE_XY, UPDATE_POSE, 2.0, 128)
    assert (pix == _update_want()).all()
    assert n == 7 + 8                                                  # steps 0..dxc of each ray blend one pixel each (:404)


# Ray D through UpdateHoleMap (:496-534): same map and pose (Scale 2, px = py = 8.5, c = 2, s = 0), HoleWidth 5.0 m, Quality 128
#   point (4.5, 5): x2p = 9, y2p = 10; xp = (int)17.5 = 17, yp = (int)18.5 = 18; dist = sqrt(81 + 100) = 13.4536;
#                   add = 5.0*2/2/13.4536 = 0.37165 -> x2p = 12.3448, y2p = 13.7165 -> x2 = (int)20.84 = 20, y2 = (int)22.22 = 22   == ray D
#   (every truncation is at least 0.16 away from an integer: no rounding detail of the float arithmetic can move it)
UPDATE_XY_D = np.array([[4.5, 5.0]], np.float32)


def test_hand_update_ray_d_c_oracle(oc):
    pix = np.full(SIZE * SIZE, FRESH, np.uint16)
    n = oc.update_holemap(pix, SIZE, 2.0, UPDATE_XY_D, UPDATE_POSE, 5.0, 128)
    assert (pix == _expect(RAY_D["want"])).all() and n == 8


def test_hand_update_ray_d_numpy_oracle(npo):
    pix = np.full(SIZE * SIZE, FRESH, np.uint16)
    pxcs = npo.poses_to_pxcs(UPDATE_POSE[None], 2.0)[0]
    npo.update_holemap_pxcs(pix, SIZE, 2.0, UPDATE_XY_D, pxcs, 5.0, 128)
    assert (pix == _expect(RAY_D["want"])).all()


@pytest.mark.gpu
def test_hand_update_ray_d_hip():
    import slam.net_amd.coreslam as cs
    ctx = cs.Context(0)
    dev = cs.CoreSlamDevice(ctx, 8.0, SIZE, 16)
    dev.set_scan(UPDATE_XY_D)
    dev.update_holemap(UPDATE_POSE, 5.0, 128)
    assert dev.last_holemap_pixels == 8
    assert (dev.holemap_download() == _expect(RAY_D["want"])).all()
    dev.close()
    ctx.close()


@pytest.mark.gpu
def test_hand_update_hip():
    import slam.net_amd.coreslam as cs
    ctx = cs.Context(0)
    dev = cs.CoreSlamDevice(ctx, 8.0, SIZE, 16)
    assert dev.hole_scale == 2.0
    dev.set_scan(UPDATE_XY)
    dev.update_holemap(UPDATE_POSE, 2.0, 128)
    assert dev.last_holemap_pixels == 15
    assert (dev.holemap_download() == _update_want()).all()
    dev.close()
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# ScanSegmentsToCloud (:187-207) with TWO segments of different poses (the case the round-5 verdict named).
#   segments.Last().Pose is the odometry pose (:719): odo = (10, 20, 0.25)
#   segment A, Pose (10.5, 20.25, 0.75):  :194 pose = A - odo = (0.5, 0.25, 0.5)   (all exact in binary32)
#       ray (Angle -0.5, Radius 2):   Angle + pose.Z = 0   -> cos 1, sin 0 -> hit = (0.5 + 2*1, 0.25 + 2*0) = (2.5, 0.25)
#       ray (Angle  0.5, Radius 1.5): Angle + pose.Z = 1.0 -> hit = (0.5 + 1.5*cos(1), 0.25 + 1.5*sin(1))
#   segment B (the last), Pose = odo: pose = (0, 0, 0)
#       ray (Angle 0, Radius 3): hit = (3, 0);   ray (Angle 1.0, Radius 4): hit = (4*cos(1), 4*sin(1))
#   cos(1) = 0.5403023058681398 lies between the binary32 neighbours 0x3F0A5140 = 0.54030227661 (2.9e-8 away) and 0x3F0A5141 = 0.54030233622
#   (3.0e-8 away): any sound cosf returns 0x3F0A5140; sin(1) = 0.8414709848078965 -> 0x3F576AA4 = 0.84147095680 (2.8e-8; the next float is
#   3.2e-8 away).  The products and sums below are formed with binary32 scalars as the calculator, one rounding per operation (:200-201).
COS1, SIN1 = np.array([0x3F0A5140], np.uint32).view(np.float32)[0], np.array([0x3F576AA4], np.uint32).view(np.float32)[0]
SEG_POSES = np.array([[10.5, 20.25, 0.75], [10.0, 20.0, 0.25]], np.float32)
SEG_START = np.array([0, 2, 4], np.int32)
SEG_RAYS = np.array([[-0.5, 2.0], [0.5, 1.5], [0.0, 3.0], [1.0, 4.0]], np.float32)
SEG_WANT = np.array([[2.5, 0.25],
                     [np.float32(0.5) + np.float32(1.5) * COS1, np.float32(0.25) + np.float32(1.5) * SIN1],
                     [3.0, 0.0],
                     [np.float32(4.0) * COS1, np.float32(4.0) * SIN1]], np.float32)


def test_hand_two_segment_cloud_c_oracle(oc):
    assert abs(float(COS1) - math.cos(1.0)) < 3e-8 and abs(float(SIN1) - math.sin(1.0)) < 3e-8
    for mode in (oc.TRIG_DET, oc.TRIG_LIBM):
        oc.set_trig_mode(mode)
        got = oc.segments_to_cloud(SEG_POSES, SEG_START, SEG_RAYS, SEG_POSES[-1])
        assert (got == SEG_WANT).all(), (mode, got)
    oc.set_trig_mode(oc.TRIG_LIBM)


def test_hand_two_segment_cloud_library():
    """the product's own ScanSegmentsToCloud (host C++, csrc/processor.hip): no device involved"""
    import ctypes as C
    import slam.net_amd.capi as capi
    out = np.zeros((4, 2), np.float32)
    rc = capi.lib().slamhip_scan_segments_to_cloud(SEG_POSES.ctypes.data_as(C.POINTER(C.c_float)), SEG_START.ctypes.data_as(C.POINTER(C.c_int32)), 2,
                                                   SEG_RAYS.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0 and (out == SEG_WANT).all(), out


# ---------------------------------------------------------------------------------------------------------------------
# Hector: one cell that a scan first marks FREE and then OCCUPIED (OccGridMap.cs:192-218).
#   8 x 8 grid, cell length 1 m -> ScaleToMap 1 (MapProperties);  robot pose (2, 3, 0), scan origin (0, 0):
#   :120-123 poseTransform = R(0) * T(2,3) * S(1) = [1 0; 0 1; 2 3];  :126-127 begin = Round((0,0)*M) = (2,3)
#   scan points, in this order:  (4, 0) -> end (6,3);   (2, 0) -> end (4,3)
#   scan 1 (currUpdateIndex 0 -> markFree 1, markOcc 2), all cells start (Value 0, UpdateIndex -1):
#     ray 1: abs_dx=4 >= abs_dy=0: Bresenham2D(4, 0, 2, +1, 0, 3*8+2=26): free 26, then i=0..2: 27, 28, 29  (abs_da - 1 = 3 steps:
#            the end point is not drawn as free);  :189 occ(30): -1 < 2, not == 1 -> Value = 0 + lo, index 2
#     ray 2: Bresenham2D(2, 0, 1, +1, 0, 26): free 26 (index 1 < 1 false: skipped), i=0: 27 (skipped);
#            occ(28): index 1 < 2 and == markFree -> Value = (lf) - lf = 0 exactly, then + lo -> lo, index 2
#     after scan 1: 26, 27, 29 = (lf, 1); 28, 30 = (lo, 2).   currUpdateIndex = 3
#   scan 2, the same scan (markFree 4, markOcc 5):
#     ray 1: free 26 -> lf + lf, 27 -> lf + lf, 28: index 2 < 4 -> lo + lf (index 4), 29 -> lf + lf;  occ(30) -> lo + lo (index 5)
#     ray 2: 26, 27 skipped;  occ(28): index 4 < 5 and == markFree -> Value = ((lo + lf) - lf) + lo, index 5
#   -- in binary32 (lo + lf) - lf need not give lo back: the expected value is formed below with exactly these three
#   IEEE operations in this order (NumPy float32 scalars as the calculator), from the implementation's own lf, lo.
def _hector_expected(lf, lo):
    lf, lo = np.float32(lf), np.float32(lo)
    want = {i: (np.float32(0.0), -1) for i in range(64)}
    want[26] = (np.float32(lf + lf), 4)
    want[27] = (np.float32(lf + lf), 4)
    want[29] = (np.float32(lf + lf), 4)
    want[30] = (np.float32(lo + lo), 5)
    want[28] = (np.float32(np.float32(np.float32(lo + lf) - lf) + lo), 5)
    return want


HECTOR_XY = np.array([[4.0, 0.0], [2.0, 0.0]], np.float32)
HECTOR_POSE = np.array([2.0, 3.0, 0.0], np.float32)


def test_hand_hector_free_then_occupied_c_oracle(oc):
    g = oc.Grid(1.0, 8, 8)
    lf, lo = g.logodds
    assert abs(lf - (-0.40546510)) < 1e-6 and abs(lo - 2.1972246) < 1e-6      # log(0.4/0.6), log(0.9/0.1) (:24-25, :88-92)
    g.update_by_scan(HECTOR_XY, HECTOR_POSE)
    c = g.cells
    assert (c["update_index"][[26, 27, 29]] == 1).all() and (c["value"][[26, 27, 29]] == np.float32(lf)).all()
    assert (c["update_index"][[28, 30]] == 2).all() and (c["value"][[28, 30]] == np.float32(lo)).all()
    g.update_by_scan(HECTOR_XY, HECTOR_POSE)
    c = g.cells
    for i, (v, idx) in _hector_expected(lf, lo).items():
        assert c["update_index"][i] == idx and c["value"][i] == v, i
    g.close()


@pytest.mark.gpu
def test_hand_hector_free_then_occupied_hip(oc):
    import slam.net_amd.coreslam as cs
    import slam.net_amd.hector as hs
    g = oc.Grid(1.0, 8, 8)
    lf, lo = g.logodds
    g.close()
    ctx = cs.Context(0)
    rep = hs.MapRepMultiMap(1.0, (8, 8), 1, ctx=ctx)
    for _ in range(2):
        rep.UpdateByScan(hs.ScanCloud(HECTOR_XY), HECTOR_POSE)
    c = rep.Maps[0].GetCells()
    for i, (v, idx) in _hector_expected(lf, lo).items():
        assert c["update_index"][i] == idx and c["value"][i] == v, i
    rep.close()
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# CalculateDistanceSISD (CoreSLAMProcessor.cs:226-259) by hand: 8 x 8 HoleMap over 4 m (Scale 2), Pixels[y*8 + x] = 1000*y + 10*x
#   pose (1.0, 1.5, 0): px = 1.0*2 + 0.5 = 2.5, py = 1.5*2 + 0.5 = 3.5, c = cos(0)*2 = 2, s = 0          (:232-235)
#   point ( 0.5 , 0.25): x = (int)(2.5 + 2*0.5 - 0) = 3,  y = (int)(3.5 + 0 + 2*0.25) = 4   -> Pixels[4*8+3] = 4030
#   point (-2.0 , 0   ): x = (int)(2.5 - 4.0) = (int)(-1.5) = -1                            -> outside (:244): not summed, not counted
#   point (-1.2 , 0   ): x = (int)(2.5 - 2.4000001) = (int)0.0999999 = 0, y = (int)3.5 = 3  -> Pixels[3*8+0] = 3000
#   point (-1.3 , 0   ): x = (int)(2.5 - 2.5999999) = (int)(-0.0999999) = 0 (truncation TOWARDS ZERO: in the map), y = 3 -> 3000
#   sum = 10030, nb_points = 3 > 0  ->  (int)((10030 * 1024) / cloud.Points.Count) = 10 270 720 / 4 = 2 567 680   (divides by ALL points, :253)
DIST_PIX = (1000 * np.arange(8)[:, None] + 10 * np.arange(8)[None, :]).astype(np.uint16).reshape(-1)
DIST_XY = np.array([[0.5, 0.25], [-2.0, 0.0], [-1.2, 0.0], [-1.3, 0.0]], np.float32)
DIST_PXCS = np.array([[2.5, 3.5, 2.0, 0.0]], np.float32)
DIST_WANT = 2567680


def test_hand_distance_c_oracle(oc):
    d, bi, bd = oc.distance_batch_pxcs(DIST_PIX, 8, DIST_XY, DIST_PXCS)
    assert d[0] == DIST_WANT and (bi, bd) == (0, DIST_WANT)
    assert (oc.pose_to_pxcs([1.0, 1.5, 0.0], 2.0) == DIST_PXCS[0]).all()


@pytest.mark.gpu
def test_hand_distance_hip():
    import slam.net_amd.coreslam as cs
    ctx = cs.Context(0)
    dev = cs.CoreSlamDevice(ctx, 4.0, 8, 8)
    dev.holemap_upload(DIST_PIX)
    dev.set_scan(DIST_XY)
    d, bi, bd = dev.distance_pxcs(DIST_PXCS)
    assert d[0] == DIST_WANT and bd == DIST_WANT
    d2, _, _ = dev.distance_poses(np.array([[1.0, 1.5, 0.0]], np.float32))
    assert d2[0] == DIST_WANT
    dev.close()
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# UpdateObstacleMap / DrawLaserRayOnObstacleMap (:456-490, :540-593) by hand: 8 x 8 ObstacleMap over 4 m (Scale 2), MaxObstacleHits 10
#   pose (0.25, 0.25, 0): px = py = 0.25*2 + 0.5 = 1.0, c = 2, s = 0; x1 = y1 = 1
#   two identical points (2.0, 1.0): x2 = (int)(1 + 2*2.0) = 5, y2 = (int)(1 + 2*1.0) = 3
#   ray (1,1) -> (5,3): dx = 4 sx = 1 dy = 2 sy = 1, err = (dx > dy ? dx : -dy) / 2 = 2
#     (1,1) noHit; e2 = 2:  2 > -4 -> err = 0, x = 2;   2 < 2 false
#     (2,1) noHit; e2 = 0:  0 > -4 -> err = -2, x = 3;  0 < 2 -> err = 2, y = 2
#     (3,2) noHit; e2 = 2:          err = 0, x = 4;     2 < 2 false
#     (4,2) noHit; e2 = 0:          err = -2, x = 5;    0 < 2 -> err = 2, y = 3
#     (5,3) == end: Pixels[3,5] < 10 ? ++ : nothing
#   start map: -5 everywhere (UnmappedObstacleHits), but [y=1,x=2] = 3, [2,3] = 0, [3,5] = 9
#   first ray: [3,5] 9 -> 10;  second ray (same cells): [3,5] = 10 is not < 10: stays (:474-477, "k hits saturate")
#   decay over the noHit cells (:576-592): [1,1] -5 -> -4;  [1,2] 3 -> 2;  [2,3] 0 stays 0;  [2,4] -5 -> -4;  the end cell is not noHit: 10
def _obst_start():
    m = np.full((8, 8), -5, np.int8)
    m[1, 2] = 3; m[2, 3] = 0; m[3, 5] = 9
    return m


def _obst_want():
    m = np.full((8, 8), -5, np.int8)
    m[1, 1] = -4; m[1, 2] = 2; m[2, 3] = 0; m[2, 4] = -4; m[3, 5] = 10
    return m


OBST_XY = np.array([[2.0, 1.0], [2.0, 1.0]], np.float32)
OBST_POSE = np.array([0.25, 0.25, 0.0], np.float32)


def test_hand_obstaclemap_c_oracle(oc):
    m = _obst_start()
    oc.update_obstaclemap(m, 8, 2.0, OBST_XY, OBST_POSE, 10)
    assert (m == _obst_want()).all(), m


@pytest.mark.gpu
def test_hand_obstaclemap_hip():
    import slam.net_amd.coreslam as cs
    ctx = cs.Context(0)
    dev = cs.CoreSlamDevice(ctx, 4.0, 8, 8)
    assert dev.obst_scale == 2.0
    dev.obstaclemap_upload(_obst_start())
    dev.set_scan(OBST_XY)
    dev.update_obstaclemap(OBST_POSE, 10)
    assert (dev.obstaclemap_download() == _obst_want()).all()
    dev.close()
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# GetCompleteHessianDerivs + InterpMapValueWithDerivatives (HectorSLAM/Matcher/ScanMatcher.cs:135-204, :206-249) by hand.
# A 32 x 32 grid with CellLength 1 (ScaleToMap 1, Limits = Dimensions - 2 = 30: MapProperties.cs:42).  Cells hold Value 0
# -- GetCachedProbability = exp(0) / (exp(0) + 1) = 0.5 (OccGridMap.cs:101-102) -- except three cells with Value 50:
# exp(50) = 5.18e21, and 5.18e21 + 1 rounds to 5.18e21 in binary32, so their probability is exactly 1.0.
# Pose (10, 20, 0) in map coordinates: transform = rotation(0) * translation(10, 20) * scale(1)  (:139-142), sinRot = 0,
# cosRot = 1 (:145-146); a scan point (x, y) lands at map (10 + x, 20 + y).  Every number below is a dyadic fraction: the
# sums are exact in binary32 in any order, so this known answer does not depend on how the terms are grouped.
#
#   p1 = (2.25, 1.5) -> map (12.25, 21.5): indMin (12, 21), factors (0.25, 0.5), xFacInv 0.75, yFacInv 0.5      (:222-242)
#        cell (13, 21) holds 50: intensities [0.5, 1.0, 0.5, 0.5]; dx1 = -0.5, dx2 = 0, dy1 = 0, dy2 = 0.5        (:230-239)
#        P  = ((0.5*0.75 + 1.0*0.25) * 0.5) + ((0.5*0.75 + 0.5*0.25) * 0.5) = 0.3125 + 0.25 = 0.5625             (:245-246)
#        gx = -((-0.5*0.75) + (0*0.25)) = 0.375   -- the X-factors weight dx1, dx2 (:247) --
#        gy = -((0*0.5) + (0.5*0.5)) = -0.25      -- and the Y-factors dy1, dy2 (:248)
#        funVal = 1 - 0.5625 = 0.4375 (:164); rotDeriv = (-0*2.25 - 1*1.5)*0.375 + (1*2.25 - 0*1.5)*(-0.25) = -1.125 (:169-170)
#        dTr += (0.375*0.4375, -0.25*0.4375, -1.125*0.4375) = (0.1640625, -0.109375, -0.4921875)                   (:166-172)
#        H11 += 0.140625, H22 += 0.0625, H33 += 1.265625, H12 += -0.09375, H13 += -0.421875, H23 += 0.28125       (:174-180)
#   p2 = (-3.5, 0.75) -> map (6.5, 20.75): indMin (6, 20), factors (0.5, 0.75); cell (6, 21) holds 50: [0.5, 0.5, 1.0, 0.5]
#        dx1 = 0, dx2 = 0.5, dy1 = -0.5, dy2 = 0;  P = (0.5*0.25) + ((1.0*0.5 + 0.5*0.5)*0.75) = 0.125 + 0.5625 = 0.6875
#        gx = -(0 + 0.5*0.5) = -0.25, gy = -((-0.5*0.25) + 0) = 0.125, funVal = 0.3125
#        rotDeriv = (-0.75)*(-0.25) + (-3.5)*0.125 = 0.1875 - 0.4375 = -0.25
#        dTr += (-0.078125, 0.0390625, -0.078125);  H11 += 0.0625, H22 += 0.015625, H33 += 0.0625, H12 += -0.03125,
#        H13 += 0.0625, H23 += -0.03125
#   p3 = (0.5, -2.25) -> map (10.5, 17.75): four cells with 0.5: P = 0.5, gx = gy = -0 -- contributes (signed) zeros only
#   p4 = (20, -14.5) -> map (30.0, 5.5): X = 30 is NOT > Limits.X = 30, the point is inside (MapProperties.cs:83-87); indMin
#        (30, 5), factors (0, 0.5); cell (31, 5) holds 50: [0.5, 1.0, 0.5, 0.5]; dx1 = -0.5, dx2 = 0, dy1 = 0, dy2 = 0.5
#        P = ((0.5*1 + 1.0*0)*0.5) + ((0.5*1 + 0.5*0)*0.5) = 0.5;  gx = -((-0.5*1) + 0) = 0.5;  gy = -(0 + 0.5*0.5) = -0.25
#        funVal = 0.5; rotDeriv = (14.5)*0.5 + (20)*(-0.25) = 7.25 - 5 = 2.25
#        dTr += (0.25, -0.125, 1.125);  H11 += 0.25, H22 += 0.0625, H33 += 5.0625, H12 += -0.125, H13 += 1.125, H23 += -0.5625
#   p5 = (20.25, -14.5) -> map (30.25, 5.5): 30.25 > 30: out of the map, Vector3.Zero (:211-214): funVal 1, all products 0
#   totals: dTr = (0.3359375, -0.1953125, 0.5546875)
#           H11 = 0.453125, H22 = 0.140625, H33 = 6.390625, H12 = H21 = -0.25, H13 = H31 = 0.765625, H23 = H32 = -0.3125
HESS_XY = np.array([[2.25, 1.5], [-3.5, 0.75], [0.5, -2.25], [20.0, -14.5], [20.25, -14.5]], np.float32)
HESS_POSE = np.array([10.0, 20.0, 0.0], np.float32)
HESS_CELLS_50 = [21 * 32 + 13, 21 * 32 + 6, 5 * 32 + 31]
HESS_H = np.array([[0.453125, -0.25, 0.765625], [-0.25, 0.140625, -0.3125], [0.765625, -0.3125, 6.390625]], np.float32)
HESS_DTR = np.array([0.3359375, -0.1953125, 0.5546875], np.float32)


def test_hand_hessian_c_oracle(oc):
    g = oc.Grid(1.0, 32, 32)
    cells = g.cells
    cells["value"][HESS_CELLS_50] = 50.0
    assert g.prob(0) == 0.5 and g.prob(HESS_CELLS_50[0]) == 1.0
    assert (g.interp(12.25, 21.5) == np.array([0.5625, 0.375, -0.25], np.float32)).all()
    assert (g.interp(6.5, 20.75) == np.array([0.6875, -0.25, 0.125], np.float32)).all()
    assert (g.interp(30.0, 5.5) == np.array([0.5, 0.5, -0.25], np.float32)).all()
    assert (g.interp(30.25, 5.5) == 0).all()
    for threads in (1, 2, 4):                                           # (the chunks' partial sums: exact in any grouping)
        H, d = g.hessian(HESS_XY, HESS_POSE, threads)
        assert (H == HESS_H).all() and (d == HESS_DTR).all(), threads
    g.close()


@pytest.mark.gpu
def test_hand_hessian_hip():
    import slam.net_amd.coreslam as cs
    import slam.net_amd.hector as hs
    ctx = cs.Context(0)
    rep = hs.MapRepMultiMap(1.0, (32, 32), 1, ctx=ctx)
    cells = rep.Maps[0].GetCells().copy()
    cells["value"][HESS_CELLS_50] = 50.0
    rep.Maps[0].SetCells(cells)
    rep.set_scan(hs.ScanCloud(HESS_XY))
    H, d = rep.Maps[0].Hessian(HESS_POSE)
    assert (H == HESS_H).all() and (d == HESS_DTR).all(), (H, d)
    rep.close()
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# One EstimateTransformationLogLh step through MatchData(OccGridMap) (ScanMatcher.cs:64-84, :93-125) by hand, on the same kind
# of grid (32 x 32, CellLength 1, Offset 0: mapTworld = scale(1) * translation(0, 0) is the identity, GridMap.cs:46, so world
# and map coordinates coincide), EstimateIterations = 1, hint (10, 20, 0).  Three points chosen so that H is diagonal with
# powers of two (Matrix4x4.Invert of a diagonal matrix: cofactor / determinant, exact here):
#   B = (0, 3.5) -> map (10.0, 23.5): indMin (10, 23), factors (0, 0.5); cells (10, 24) and (11, 24) hold 50:
#       intensities [0.5, 0.5, 1.0, 1.0]; dx1 = dx2 = 0 -> gx = -0; dy1 = dy2 = -0.5 -> gy = -((-0.5*0.5) + (-0.5*0.5)) = 0.5
#       P = ((0.5*1 + 0.5*0)*0.5) + ((1.0*1 + 1.0*0)*0.5) = 0.75, funVal 0.25; rotDeriv = (-3.5)*(-0) + (0)*0.5 = 0
#       dTr.Y += 0.5*0.25 = 0.125; H22 += 0.25
#   C = (0, 1) -> map (10.0, 21.0): indMin (10, 21), factors (0, 0); cell (11, 21) holds 50: [0.5, 1.0, 0.5, 0.5]
#       gx = -((0.5 - 1.0)*1 + (0)*0) = 0.5; gy = -((0.5 - 0.5)*1 + (0.5)*0) = -0; P = 0.5, funVal 0.5; rotDeriv = (-1)*0.5 = -0.5
#       dTr.X += 0.25, dTr.Z += -0.25; H11 += 0.25, H33 += 0.25, H13 += -0.25
#   D = (0, -1) -> map (10.0, 19.0): indMin (10, 19); cell (11, 19) holds 50: gx = 0.5, gy = -0, P = 0.5; rotDeriv = (+1)*0.5 = 0.5
#       dTr.X += 0.25, dTr.Z += +0.25; H11 += 0.25, H33 += 0.25, H13 += +0.25
#   H = diag(0.5, 0.25, 0.5) (M44 = 1, :202), dTr = (0.5, 0.125, 0); M11 and M22 are not zero (:97)
#   iH = diag(2, 4, 2, 1); searchDir = Vector3.Transform(dTr, iH) = (0.5*2, 0.125*4, 0*2) = (1.0, 0.5, 0) (:105); |Z| <= 0.2
#   estimate = (10, 20, 0) + (1.0, 0.5, 0) = (11.0, 20.5, 0) (:119); NormalizeAngle(0) = 0 (:76); world = map (:79)
STEP_XY = np.array([[0.0, 3.5], [0.0, 1.0], [0.0, -1.0]], np.float32)
STEP_HINT = np.array([10.0, 20.0, 0.0], np.float32)
STEP_CELLS_50 = [24 * 32 + 10, 24 * 32 + 11, 21 * 32 + 11, 19 * 32 + 11]
STEP_H = np.diag([0.5, 0.25, 0.5]).astype(np.float32)
STEP_DTR = np.array([0.5, 0.125, 0.0], np.float32)
STEP_WANT = np.array([11.0, 20.5, 0.0], np.float32)


def test_hand_estimate_step_c_oracle(oc):
    g = oc.Grid(1.0, 32, 32)
    g.cells["value"][STEP_CELLS_50] = 50.0
    H, d = g.hessian(STEP_XY, STEP_HINT, 1)
    assert (H == STEP_H).all() and (d == STEP_DTR).all()
    assert (g.match(STEP_XY, STEP_HINT, iterations=1) == STEP_WANT).all()
    g.close()


@pytest.mark.gpu
def test_hand_estimate_step_hip():
    import slam.net_amd.coreslam as cs
    import slam.net_amd.hector as hs
    ctx = cs.Context(0)
    rep = hs.MapRepMultiMap(1.0, (32, 32), 1, ctx=ctx)
    cells = rep.Maps[0].GetCells().copy()
    cells["value"][STEP_CELLS_50] = 50.0
    rep.Maps[0].SetCells(cells)
    rep.Maps[0].EstimateIterations = 1
    got = hs.ScanMatcher(1).MatchData(rep.Maps[0], hs.ScanCloud(STEP_XY), STEP_HINT)
    assert (got == STEP_WANT).all(), got
    rep.close()
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# One EstimateTransformationLogLh step whose H has OFF-DIAGONAL entries, so that Matrix4x4.Invert's cofactors -- not only its
# diagonal -- decide the step (ScanMatcher.cs:93-125), with the clamp of :107-111 and the float rounding of NormalizeAngle
# (:76, MathEx.cs:116-138) behind it.  Same grid as above (32 x 32, CellLength 1: map = world), EstimateIterations = 1, hint
# (10, 20, 0): sinRot = 0, cosRot = 1 (:145-146), rotDeriv = -y * gx + x * gy (:169-170).
#   B = (0, 3.5): as above -- dTr.Y += 0.125, H22 += 0.25, gx = -0, rotDeriv = 0
#   P1 = (0, 0) -> map (10, 20), factors (0, 0); cell (11, 20) holds 50: intensities [0.5, 1.0, 0.5, 0.5]
#       gx = -((0.5 - 1.0)*1 + 0*0) = 0.5; gy = -(0*1 + 0.5*0) = -0; P = 0.5, funVal 0.5; rotDeriv = -0*0.5 + 0*(-0) = -0
#       dTr.X += 0.25; H11 += 0.25 (nothing else: every other product has a zero factor)
#   P2 = (4, 1) -> map (14, 21), factors (0, 0); cells (14, 21) and (14, 22) hold 50: intensities [1.0, 0.5, 1.0, 0.5]
#       gx = -((1.0 - 0.5)*1 + (1.0 - 0.5)*0) = -0.5; gy = -((1.0 - 1.0)*1 + (0.5 - 0.5)*0) = -0; P = 1.0, funVal = 0
#       rotDeriv = (-1)*(-0.5) + 4*(-0) = 0.5;  dTr += 0;  H11 += 0.25, H33 += 0.25, H13 += (-0.5)(0.5) = -0.25
#   H = [[0.5, 0, -0.25], [0, 0.25, 0], [-0.25, 0, 0.25]] (M44 = 1, :202), dTr = (0.25, 0.125, 0); M11, M22 != 0 (:97)
#   Matrix4x4.Invert (cofactor expansion; a..p row by row: a=0.5 c=-0.25 f=0.25 i=-0.25 k=0.25 p=1, the rest 0):
#       kp_lo = k*p - l*o = 0.25, ip_lm = i*p - l*m = -0.25, the other 2 x 2 minors of rows 3, 4 are (+-)0
#       a11 = f*kp_lo = 0.0625;  a13 = -(f*ip_lm) = 0.0625;  a12 = a14 = (+-)0
#       det = a*a11 + c*a13 = 0.03125 - 0.015625 = 0.015625 = 2^-6  (every product and sum exact in binary32);  invDet = 64
#       M11 = a11*64 = 4;  M31 = a13*64 = 4;  M22 = (a*kp_lo - c*ip_lm)*64 = (0.125 - 0.0625)*64 = 4
#       M13 = (b*(g*p - h*o) - c*(f*p - h*n) + d*(f*o - g*n))*64 = (0 + 0.25*0.25 + 0)*64 = 4;  M33 = (a*(f*p - h*n))*64 = 0.125*64 = 8
#       M12, M21, M23, M32 = (+-)0
#   iH = [[4, 0, 4], [0, 4, 0], [4, 0, 8]] -- a diagonal-only inverse (1/H11, 1/H22, 1/H33) = diag(2, 4, 4) would give the step (0.5, 0.5, 0)
#   searchDir = Vector3.Transform(dTr, iH) (:105): X = 0.25*4 + 0.125*0 + 0*4 = 1.0;  Y = 0.125*4 = 0.5;  Z = 0.25*4 + 0 + 0*8 = 1.0
#   :107-111 Z > 0.2 -> Z = 0.2f;  estimate = (11, 20.5, 0.2f) (:119)
#   :76 NormalizeAngle(0.2f), pi2 = 3.14159274f * 2 = 6.28318548f = K * 2^-21 (binade [4, 8): ulp 2^-21):
#       fmodf(0.2f, pi2) = 0.2f = 419430.40625 * 2^-21;  0.2f + pi2 rounds to (K + 419430) * 2^-21;  fmodf(that, pi2) = 419430 * 2^-21
#       = 0.19999980926513671875 = 0x3E4CCCC0 (not > pi): the pose's angle -- 13 ulp below 0.2f
OFFD_XY = np.array([[0.0, 3.5], [0.0, 0.0], [4.0, 1.0]], np.float32)
OFFD_HINT = np.array([10.0, 20.0, 0.0], np.float32)
OFFD_CELLS_50 = [24 * 32 + 10, 24 * 32 + 11, 20 * 32 + 11, 21 * 32 + 14, 22 * 32 + 14]
OFFD_H = np.array([[0.5, 0.0, -0.25], [0.0, 0.25, 0.0], [-0.25, 0.0, 0.25]], np.float32)
OFFD_DTR = np.array([0.25, 0.125, 0.0], np.float32)
OFFD_WANT = np.array([11.0, 20.5, np.array([0x3E4CCCC0], np.uint32).view(np.float32)[0]], np.float32)


def test_hand_offdiagonal_step_c_oracle(oc):
    g = oc.Grid(1.0, 32, 32)
    g.cells["value"][OFFD_CELLS_50] = 50.0
    H, d = g.hessian(OFFD_XY, OFFD_HINT, 1)
    assert (H == OFFD_H).all() and (d == OFFD_DTR).all(), (H, d)
    got = g.match(OFFD_XY, OFFD_HINT, iterations=1)
    assert (got.view(np.uint32) == OFFD_WANT.view(np.uint32)).all(), got
    g.close()


def test_hand_offdiagonal_hessian_numpy_oracle(npo):
    ng = npo.NpGrid(1.0, 32, 32)
    ng.value[OFFD_CELLS_50] = 50.0
    H, d = ng.hessian(OFFD_XY, OFFD_HINT, 1)
    assert (H == OFFD_H).all() and (d == OFFD_DTR).all(), (H, d)


@pytest.mark.gpu
def test_hand_offdiagonal_step_hip():
    import slam.net_amd.coreslam as cs
    import slam.net_amd.hector as hs
    ctx = cs.Context(0)
    rep = hs.MapRepMultiMap(1.0, (32, 32), 1, ctx=ctx)
    cells = rep.Maps[0].GetCells().copy()
    cells["value"][OFFD_CELLS_50] = 50.0
    rep.Maps[0].SetCells(cells)
    rep.set_scan(hs.ScanCloud(OFFD_XY))
    H, d = rep.Maps[0].Hessian(OFFD_HINT)
    assert (H == OFFD_H).all() and (d == OFFD_DTR).all(), (H, d)
    rep.Maps[0].EstimateIterations = 1
    got = hs.ScanMatcher(1).MatchData(rep.Maps[0], hs.ScanCloud(OFFD_XY), OFFD_HINT)
    assert (np.asarray(got, np.float32).view(np.uint32) == OFFD_WANT.view(np.uint32)).all(), got
    rep.close()
    ctx.close()
