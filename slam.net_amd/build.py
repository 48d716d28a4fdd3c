"""hipcc build recipe for libslamhip.so (gfx950 only, in-tree so the .so travels with gpurun snapshots).

Flags that are part of the numerical contract:
  -ffp-contract=off   the reference's pixel coordinates are separate binary32 mul/add roundings
                      (CoreSLAMProcessor.cs:240-241); an FMA would change map cells
  no fast-math; IEEE divide / sqrt (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt)
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libslamhip.so")
SOURCES = ["context.hip", "distance.hip", "holemap.hip", "obstacle.hip", "coreslam.hip", "processor.hip",
           "hector.hip", "group.hip"]
HEADERS = ["common.h", "cs_internal.h", "det_trig.h", "m3x2.h", "raster.h", "obstacle_dev.h", "k1_pieces.inc", os.path.join("..", "..", "include", "slamhip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-slp-vectorize",      # v_pk_*_f32 runs at half rate on CDNA4 and computed unused lanes in K1

         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def extra_defs():
    """Compile-time tunables (measured defaults live in the sources)."""
    d = []
    if os.environ.get("SLAMHIP_K1_TIMES"):      # developer build: per-workgroup phase stamps in the fused K1 kernel
        d.append("-DK1_TIMES=1")
    if os.environ.get("SLAMHIP_K4_TIMES"):      # developer build: phase stamps in the Hector matcher
        d.append("-DK4_TIMES=1")
    if os.environ.get("SLAMHIP_K2_TIMES"):      # developer build: per-workgroup phase stamps in the K2 pixel kernel
        d.append("-DK2_TIMES=1")
    if os.environ.get("SLAMHIP_K5_TIMES"):      # developer build: per-workgroup phase stamps in the Hector cell kernel
        d.append("-DK5_TIMES=1")
    if os.environ.get("SLAMHIP_K1_DMA0"):       # developer A/B: 0 = the first tile through staging registers like the later ones
        d.append("-DK1_DMA0=%s" % os.environ["SLAMHIP_K1_DMA0"])
    # (the wrong-results experiment builds -DK1_EXP=n / -DK2_EXP=n are made as side variants only: tools/build_variant.py WORK <tag> -DK1_EXP=n,
    # selected with SLAMHIP_LIB -- never as slam.net_amd/libslamhip.so)
    if os.environ.get("SLAMHIP_K2_LDS_RAYS"):   # developer experiment: rays of the K2 pixel kernel's LDS table
        d.append("-DK2_LDS_RAYS=%s" % os.environ["SLAMHIP_K2_LDS_RAYS"])
    return d


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libslamhip cannot be built (there is no CPU fallback)")


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return OUT
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        spath = os.path.join(CSRC, src)
        hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(spath)
                and os.path.getmtime(obj) > hdr_t and os.path.getmtime(obj) > os.path.getmtime(os.path.abspath(__file__))):
            continue
        cmd = [cc] + FLAGS + extra_defs() + ["-c", spath, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl", "-lpthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
