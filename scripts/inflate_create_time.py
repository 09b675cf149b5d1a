import time, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganon_amd import hip
import numpy as np
hip.load_library()
for size in (600_000_000, 600_000_000, 100_000_000):
    t0 = time.time()
    z = hip.HipInflate(size)
    t1 = time.time()
    z.close()
    t2 = time.time()
    print("compressed", size, "create %.3f s, destroy %.3f s" % (t1 - t0, t2 - t1))
