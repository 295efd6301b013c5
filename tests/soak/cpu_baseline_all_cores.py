import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle
n = oracle.max_threads()
print("host threads available:", n, "affinity:", len(os.sched_getaffinity(0)))
for th in (1, 16):
    oracle.set_threads(th)
    oracle.build_grid((360, 180, 1))
    t0 = time.perf_counter(); oracle.build_grid((1440, 720, 1)); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); oracle.build_grid((3600, 1800, 1)); t2 = time.perf_counter() - t0
    size, halo = (3600, 64, 75), (4, 4, 4)
    fs = [np.random.default_rng(i).uniform(-1, 1, (83, 72, 3608)) for i in range(4)]
    specs = [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1)]
    for f, (x, y, s) in zip(fs, specs): oracle.zipper_fill(f, x, y, s, size, halo)
    t0 = time.perf_counter()
    for _ in range(5):
        for f, (x, y, s) in zip(fs, specs): oracle.zipper_fill(f, x, y, s, size, halo)
    tz = (time.perf_counter() - t0) / 5
    print(f"threads={th}: 1/4deg build {t1:.3f}s = {1440*720/t1:.3e} cells/s; 1/10deg build {t2:.3f}s = {3600*1800/t2:.3e} cells/s; zipper {tz*1e3:.2f} ms = {73.44e6/tz/1e9:.1f} GB/s")
