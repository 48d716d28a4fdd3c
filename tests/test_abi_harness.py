"""The C-ABI driven by a caller that is not Python: tests/abi_harness.c is compiled with gcc, dlopen()s libslamhip.so the way
a P/Invoke binding does (entry points by name, prototypes restated on the caller's side, caller-owned malloc buffers) and
replays one golden fixture each for K1 (distance batch + search), K2 (HoleMap update, ten scans), K3 (ObstacleMap update), K5 (Hector
grid update) and K4 (a match on that grid), and drives the multi-GPU entry points on a group of one GPU.
SURVEY.md sec.8b / H9: the C# shim, the ctypes harness and a native harness all go through the same extern "C" symbols."""
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
SRC = os.path.join(HERE, "abi_harness.c")


def build_harness(tmp):
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    exe = os.path.join(tmp, "abi_harness")
    subprocess.check_call([cc, "-O1", "-Wall", "-Werror", "-o", exe, SRC, "-ldl", "-lm"])
    return exe


def lib_path():
    import slam.net_amd.capi as capi
    capi.lib()                               # (builds it when stale; raises when it cannot be built)
    return os.path.join(ROOT, "slam.net_amd", "libslamhip.so")


def test_harness_builds_and_resolves_every_symbol(tmp_path):
    """CPU leg: the harness compiles warning-free and finds every entry point it binds by name (no compute, no GPU)."""
    exe = build_harness(str(tmp_path))
    r = subprocess.run([exe, lib_path(), "--symbols-only"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    assert b"symbols ok" in r.stdout


def write_fixtures(d):
    g = np.load(os.path.join(GOLD, "cs_distance_256_r360_k256.npz"))
    size = int(g["size"])
    with open(os.path.join(d, "k1.meta"), "w") as f:
        f.write("%d %r %d %d\n" % (size, 40.0, g["xy"].shape[0], g["pxcs"].shape[0]))
    assert np.float32(size) / np.float32(40.0) == g["scale"]
    g["pixels"].astype("<u2").tofile(os.path.join(d, "k1_pixels.u16"))
    g["xy"].astype("<f4").tofile(os.path.join(d, "k1_xy.f32"))
    g["pxcs"].astype("<f4").tofile(os.path.join(d, "k1_pxcs.f32"))
    g["base"].astype("<f4").tofile(os.path.join(d, "k1_base.f32"))
    g["offs"].astype("<f4").tofile(os.path.join(d, "k1_offs.f32"))
    g["dist"].astype("<i4").tofile(os.path.join(d, "k1_dist.i32"))

    g = np.load(os.path.join(GOLD, "cs_holemap_256_r360_hw2.npz"))
    size = int(g["size"])
    assert np.float32(size) / np.float32(40.0) == g["scale"]
    with open(os.path.join(d, "k2.meta"), "w") as f:
        f.write("%d %r %d %d %r %d\n" % (size, 40.0, g["xy"].shape[1], g["xy"].shape[0], float(g["hole_width"]), int(g["quality"])))
    g["xy"].astype("<f4").tofile(os.path.join(d, "k2_xy.f32"))
    g["pxcs"].astype("<f4").tofile(os.path.join(d, "k2_pxcs.f32"))
    g["after1"].astype("<u2").tofile(os.path.join(d, "k2_after1.u16"))
    g["after_all"].astype("<u2").tofile(os.path.join(d, "k2_after_all.u16"))
    g["counts"].astype("<i8").tofile(os.path.join(d, "k2_counts.i64"))

    g = np.load(os.path.join(GOLD, "cs_obstacle_64_r360.npz"))
    osize = int(g["size"])
    with open(os.path.join(d, "k3.meta"), "w") as f:      # (the ObstacleMap's scale is obst_size / physical: 64 cells over 40 m)
        f.write("%d %r %d %d %d\n" % (osize, 40.0, g["xy"].shape[1], g["xy"].shape[0], int(g["max_hits"])))
    assert np.float32(osize) / np.float32(40.0) == g["scale"]
    g["xy"].astype("<f4").tofile(os.path.join(d, "k3_xy.f32"))
    g["pxcs"].astype("<f4").tofile(os.path.join(d, "k3_pxcs.f32"))
    g["after1"].astype("i1").tofile(os.path.join(d, "k3_after1.i8"))
    g["after_all"].astype("i1").tofile(os.path.join(d, "k3_after_all.i8"))

    g = np.load(os.path.join(GOLD, "hs_grid_200_r180.npz"))
    with open(os.path.join(d, "k5.meta"), "w") as f:      # (the cell length as its binary32 bit pattern: exact through a text file)
        f.write("%d %d %d %d\n" % (int(g["side"]), int(np.float32(g["cell"]).view(np.uint32)), g["xy"].shape[1], g["xy"].shape[0]))
    g["xy"].astype("<f4").tofile(os.path.join(d, "k5_xy.f32"))
    g["poses"].astype("<f4").tofile(os.path.join(d, "k5_poses.f32"))
    g["value"].astype("<f4").tofile(os.path.join(d, "k5_value.f32"))
    g["upd"].astype("<i4").tofile(os.path.join(d, "k5_upd.i32"))
    # K4: a match on the grid those scans built -- the expected pose from the CPU checker (test infrastructure), run here
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c as oc
    oc.set_trig_mode(oc.TRIG_DET)
    side, cell = int(g["side"]), float(g["cell"])
    grid = oc.Grid(cell, side, side)
    for xy, pose in zip(g["xy"], g["poses"]):
        grid.update_by_scan(xy, pose)
    hint = (g["poses"][-1] + np.array([0.06, -0.05, 0.02], np.float32)).astype(np.float32)
    want = grid.match(g["match_xy"], hint, iterations=3, n_threads=1)
    with open(os.path.join(d, "k4.meta"), "w") as f:
        f.write("%d\n" % g["match_xy"].shape[0])
    g["match_xy"].astype("<f4").tofile(os.path.join(d, "k4_xy.f32"))
    hint.astype("<f4").tofile(os.path.join(d, "k4_hint.f32"))
    np.asarray(want, "<f4").tofile(os.path.join(d, "k4_pose.f32"))


@pytest.mark.gpu
def test_native_caller_replays_golden_fixtures(tmp_path):
    exe = build_harness(str(tmp_path))
    d = str(tmp_path / "fx")
    os.makedirs(d)
    write_fixtures(d)
    r = subprocess.run([exe, lib_path(), d], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out
    assert "k1 ok" in out and "k2 ok" in out and "k3 ok" in out and "k4 ok" in out and "k5 ok" in out and "all golden replays bit-exact" in out


@pytest.mark.gpu
def test_native_caller_drives_a_group_of_one_gpu(tmp_path):
    """--group: the multi-GPU entry points (slamhip_group_*: block-sharded search, the exchange of the packed keys, the fused scan on
    every GPU, the replica check) on a group of ONE GPU, from a caller that is not Python -- the path the driver's 8-GPU box runs
    first (CoreSLAMProcessor.cs:674-710 across devices instead of threads)."""
    exe = build_harness(str(tmp_path))
    d = str(tmp_path / "fx")
    os.makedirs(d)
    write_fixtures(d)
    r = subprocess.run([exe, lib_path(), "--group", d], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "group ok" in out, out


@pytest.mark.gpu
def test_native_caller_times_processor_update(tmp_path):
    """--bench-proc: CoreSLAMProcessor.Update from the native caller (a short run: the mode works and tracks the room)."""
    exe = build_harness(str(tmp_path))
    r = subprocess.run([exe, lib_path(), "--bench-proc", "512", "360", "2049", "40"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "proc_us_per_scan" in out, out
    us = float(out.split("proc_us_per_scan")[1].split()[0])
    assert 1.0 < us < 5000.0, out


@pytest.mark.gpu
def test_native_caller_times_hector_processor_update(tmp_path):
    """--bench-hsproc: HectorSLAMProcessor.Update from the native caller (a short run: the mode works, every scan updates the
    grids and the match tracks the room: the last match lies where the lap's last scan was taken)."""
    exe = build_harness(str(tmp_path))
    r = subprocess.run([exe, lib_path(), "--bench-hsproc", "512", "3", "360", "40"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "hsproc_us_per_scan" in out, out
    us = float(out.split("hsproc_us_per_scan")[1].split()[0])
    assert 1.0 < us < 5000.0, out
    assert "280 of 280 scans updated the grids" in out, out
    mx, my, mth = [float(v) for v in out.split("last match")[1].replace(")", " ").split()[:3]]
    # the timed loop ends with scan kk = 7 * 40 - 1 of the lap: j = 12 + kk % 52
    j = 12 + (7 * 40 - 1) % 52
    assert abs(mx - (20.0 + 0.04 * j)) < 0.08 and abs(my - (20.0 + 0.015 * j)) < 0.08 and abs(mth - 0.004 * j) < 0.02, out
