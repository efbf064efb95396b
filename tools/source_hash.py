"""sha1 (12 hex digits) over the kernel sources libppk.so is built from: ties a committed profile to the code it
was taken from (bench.py refuses PMC traffic figures whose hash differs from the tree it runs in)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha1()
    d = os.path.join(ROOT, 'paccmann_polymer_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(d, '*.hip')) + glob.glob(os.path.join(d, '*.h')) + [os.path.join(d, 'Makefile')]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:12]


if __name__ == '__main__':
    print(source_hash())
