#!/bin/bash
# Builds paccmann_polymer_amd/libppk_timing.so: libppk.so with the per-phase cycle stamps of the persistent sequence
# kernels compiled in (PPK_SEQ_TIMING).  Use: PPK_LIB_PATH=paccmann_polymer_amd/libppk_timing.so python tools/seq_timing.py
set -e
cd "$(dirname "$0")/../paccmann_polymer_amd/csrc"
make -s -j8
mkdir -p /tmp/ppk_timing
for f in ppk_seq_persist ppk_seq_persist_bwd; do
  [ -f $f.hip ] && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -DPPK_SEQ_TIMING -c $f.hip -o /tmp/ppk_timing/$f.o
done
objs=""
for o in ppk_*.o; do
  if [ -f /tmp/ppk_timing/$o ]; then objs="$objs /tmp/ppk_timing/$o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libppk_timing.so $objs
