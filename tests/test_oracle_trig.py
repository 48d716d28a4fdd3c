"""Deterministic float trig (oracle/det_trig.c, restated in np_oracle.det_sincos and on the device)
vs this machine's libm.  MathF.Cos/Sin in the reference are the platform CRT's cosf/sinf
(CoreSLAMProcessor.cs:234-235), which no reference test pins: PARITY UNPINNED at the ulp level."""
import math

import numpy as np


def test_det_trig_c_equals_numpy(oc, npo):
    rng = np.random.default_rng(11)
    a = np.concatenate([rng.uniform(-7, 7, 20000), rng.uniform(-1000, 1000, 5000),
                        [0.0, math.pi, -math.pi, math.pi / 2, 1e-8, -1e-8, 65536.0, -65535.5]]).astype(np.float32)
    s1, c1 = oc.det_sincos_array(a)
    s2, c2 = npo.det_sincos(a)
    assert (s1 == s2).all() and (c1 == c2).all()


def test_det_trig_is_correctly_rounded(npo):
    """Agreement with the correctly rounded value (float64 libm, rounded once) must be total on a
    200k sample; agreement with float32 libm (glibc cosf/sinf, <= 0.56 ulp) is reported as a rate."""
    rng = np.random.default_rng(12)
    a = rng.uniform(-6.5, 6.5, 200000).astype(np.float32)
    s, c = npo.det_sincos(a)
    s64 = np.sin(a.astype(np.float64)).astype(np.float32)
    c64 = np.cos(a.astype(np.float64)).astype(np.float32)
    assert (s == s64).all() and (c == c64).all()


def test_det_trig_vs_libm_rate(oc):
    rng = np.random.default_rng(13)
    a = rng.uniform(-3.2, 3.2, 50000).astype(np.float32)
    s, c = oc.det_sincos_array(a)
    L = oc.lib()
    import ctypes as C
    oc.set_trig_mode(oc.TRIG_LIBM)
    sl = np.array([L.oracle_sinf(C.c_float(v)) for v in a], np.float32)
    cl = np.array([L.oracle_cosf(C.c_float(v)) for v in a], np.float32)
    rate = ((s == sl) & (c == cl)).mean()
    ulp = np.maximum(np.abs(s - sl) / np.spacing(np.abs(sl) + 1e-30), np.abs(c - cl) / np.spacing(np.abs(cl) + 1e-30))
    print("det trig == libm cosf/sinf on %.4f%% of 50000 angles; max diff %.1f ulp" % (100 * rate, ulp.max()))
    assert rate > 0.95 and ulp.max() <= 1.0     # glibc cosf/sinf are <=0.56 ulp, not correctly rounded
