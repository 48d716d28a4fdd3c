"""Developer aid: build libslamhip from the sources of a git revision (or the working tree: WORK) into slam.net_amd/build/variants/<tag>.so,
so that several versions of a kernel can be timed on ONE box in one gpurun call (SLAMHIP_LIB=... selects the library; tools/ab_k2.sh).
    python tools/build_variant.py <rev|WORK> <tag> [-DNAME=VALUE ...]"""
import os, subprocess, sys, tempfile, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import slam.net_amd.build as b
rev, tag = sys.argv[1], sys.argv[2]
defs = sys.argv[3:]
tmp = tempfile.mkdtemp(prefix="slamvar_")
os.makedirs(os.path.join(tmp, "slam.net_amd", "csrc")); os.makedirs(os.path.join(tmp, "include"))
if rev == "WORK":
    for f in os.listdir(b.CSRC): shutil.copy(os.path.join(b.CSRC, f), os.path.join(tmp, "slam.net_amd", "csrc", f))
    shutil.copy(os.path.join(root, "include", "slamhip.h"), os.path.join(tmp, "include", "slamhip.h"))
else:
    names = subprocess.check_output(["git", "-C", root, "ls-tree", "--name-only", rev, "slam.net_amd/csrc/"]).decode().split()
    for n in names:
        open(os.path.join(tmp, n), "wb").write(subprocess.check_output(["git", "-C", root, "show", "%s:%s" % (rev, n)]))
    open(os.path.join(tmp, "include", "slamhip.h"), "wb").write(subprocess.check_output(["git", "-C", root, "show", "%s:include/slamhip.h" % rev]))
outdir = os.path.join(b.HERE, "build", "variants"); os.makedirs(outdir, exist_ok=True)
procs, objs = [], []
for src in b.SOURCES:
    o = os.path.join(tmp, src.replace(".hip", ".o")); objs.append(o)
    procs.append(subprocess.Popen([b.hipcc()] + b.FLAGS + defs + ["-c", os.path.join(tmp, "slam.net_amd", "csrc", src), "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
for p in procs:
    out, _ = p.communicate()
    if p.returncode: print(out.decode(errors="replace")); sys.exit(1)
so = os.path.join(outdir, tag + ".so")
subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + ["-ldl", "-lpthread"])
shutil.rmtree(tmp)
print(so)
