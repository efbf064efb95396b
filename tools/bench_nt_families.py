import os, sys
sys.path.insert(0, '/root/repo')
sys.path.insert(0, '/root/repo/tools')
import bench_nt
from paccmann_polymer_amd import _C
for lim, name in ((0, 'nt64'), (1 << 30, 'nt32')):
    _C.lib.ppk_nt_small_grid_limit(lim)
    print(name)
    for (N, K) in ((512, 1536), (1536, 512)):
        for M in (1024, 1536, 2048, 2304, 2560, 3072, 3584, 4096):
            bench_nt.bench(M, N, K)
