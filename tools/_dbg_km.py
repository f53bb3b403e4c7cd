import sys, numpy as np
sys.path.insert(0, '.')
from graphrole_amd import kernels as K
rng = np.random.RandomState(0)
for m, k in [(600, 8), (600, 64), (5000, 8), (70000, 16), (1000000, 64), (3000000, 32)]:
    data = rng.rand(m)
    q, c, info = K.kmeans1d(K.to_device(data), k)
    print(m, k, K.to_host(info))
