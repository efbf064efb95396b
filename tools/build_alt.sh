#!/bin/bash
# Builds paccmann_polymer_amd/libppk_alt.so: libppk.so with the sequence kernels recompiled with extra -D flags, for
# same-box A/B runs (tools/ab_bench.sh).   bash tools/build_alt.sh -DPPK_POLL_SLEEP=1
set -e
cd "$(dirname "$0")/../paccmann_polymer_amd/csrc"
make -s -j8
mkdir -p /tmp/ppk_alt
for f in ppk_seq_persist ppk_seq_persist_bwd ppk_seq_ws; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function "$@" -c $f.hip -o /tmp/ppk_alt/$f.o
done
objs=""
for o in ppk_*.o; do
  if [ -f /tmp/ppk_alt/$o ]; then objs="$objs /tmp/ppk_alt/$o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libppk_alt.so $objs
