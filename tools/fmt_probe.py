import sys, os, tempfile
sys.path.insert(0, ".")
import numpy as np
from tests.test_dist_cpu import run_workers
for world in (2, 3):
    d = tempfile.mkdtemp()
    res = run_workers("gcr", world, d, timeout=240)
    print("world", world, "poisson48 formats", [res[r]["poisson48"]["format"] for r in range(world)], "poisson", [res[r]["poisson"]["format"] for r in range(world)])
d = tempfile.mkdtemp()
res = run_workers("slab", 2, d, timeout=240)
print("slab formats", [res[r]["slab"]["format"] for r in range(2)], [res[r]["slab"]["layout"] for r in range(2)])
