"""nt64 / nt32 / nt128 tile families on the same isolated launches (run once per family:
PPK_NT_FAMILY=32|64|128 python tools/bench_nt_families.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_nt
print('family', os.environ.get('PPK_NT_FAMILY', 'model'))
for (N, K) in ((512, 1536), (1536, 512)):
    for M in (512, 1024, 1536, 2048, 2304, 2560, 3072, 3584, 4096):
        bench_nt.bench(M, N, K)
