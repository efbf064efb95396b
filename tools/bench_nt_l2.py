import os, sys
sys.path.insert(0, '/root/repo/tools')
import bench_nt
print('family', os.environ.get('PPK_NT_FAMILY'))
for (M, N, K) in ((1024, 256, 1536), (2048, 256, 1536), (1024, 512, 1536), (2048, 512, 1536), (2048, 512, 512), (2048, 512, 3072), (2048, 512, 6144)):
    bench_nt.bench(M, N, K)
