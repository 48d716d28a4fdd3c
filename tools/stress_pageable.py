"""stress: whole-map transfers to / from pageable NumPy arrays that live in the brk heap (MALLOC_MMAP_THRESHOLD_ raised), with the heap
moving underneath (allocations and frees of other sizes between the calls)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam.net_amd.coreslam as cs, slam.net_amd.sim as sim
ctx = cs.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 120
segs = sim.default_field(); prng = sim.PCG32(3)
n = 0
keep = []
while time.time() < t_end:
    size = int(rng.choice([1024, 2048, 2048, 4096]))
    dev = cs.CoreSlamDevice(ctx, 40.0, size, 256)
    _, xy = sim.make_scan(segs, np.array([20, 20, 0.3], np.float32), 1080, prng)
    dev.set_scan(xy)
    for it in range(int(rng.integers(2, 8))):
        dev.update_holemap(np.array([20 + 0.1 * it, 20, 0.3], np.float32), 0.6, 50)
        a = dev.holemap_download()
        junk = [np.empty(int(rng.integers(1, 3_000_000)), np.uint8) for _ in range(int(rng.integers(0, 4)))]
        if rng.random() < 0.5: keep.append(np.empty(int(rng.integers(1, 500_000)), np.uint8))
        if len(keep) > 40: del keep[:20]
        b = dev.obstaclemap_download()
        dev.holemap_upload(a)
        offs = sim.gaussian_offsets(int(rng.choice([1023, 16383, 65535])), 0.1, 0.17, seed=int(rng.integers(1, 1 << 30)))
        dev.set_offsets(offs)
        dev.search(np.array([20, 20, 0.3], np.float32))
        c = dev.holemap_download()
        if n == 3: print('a download lives at', hex(c.ctypes.data), flush=True)
        assert (a == c).all()
        del junk, a, b, c
        n += 1
    dev.close()
print("pageable transfers: %d rounds clean" % n)
