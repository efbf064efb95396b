#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the small-batch regime (trainer.train() on bundles of 6..21 SMILES) with the
# weights-stationary sequence kernels: samples/s per launch schedule, rocprofv3 kernel stats of the default schedule,
# per-phase cycles (needs paccmann_polymer_amd/libppk_alt.so built with -DPPK_WS_TIMING).  Output: gpurun_out/wsprof/
#   bash tools/collect_ws_profiles.sh
set -e
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
W=$R/gpurun_out/wsprof; mkdir -p $W
H=$(python3 $R/tools/source_hash.py)
cd $R
{ echo "# source_hash $H"; echo "# python tools/small_batch_bench.py  (trainer.train() on bundles of 6..21 SMILES, bench architecture)";
  timeout -k 10 300 python3 -u tools/small_batch_bench.py 2>/dev/null; } > $W/small_batch.txt
{ echo; echo "# python tools/ws_check.py B 100  (encoder forward + backward: per-step launches / row-block persistent / weights-stationary)";
  for B in 6 16 21 32; do timeout -k 10 200 python3 -u tools/ws_check.py $B 100 2>/dev/null | grep -E "^mode|forward kernels" ; done; } >> $W/small_batch.txt
if [ -f paccmann_polymer_amd/libppk_alt.so ]; then
  { echo "# source_hash $H   (PPK_WS_TIMING build of the same sources: bash tools/build_alt.sh -DPPK_WS_TIMING)";
    for B in 16 21; do PPK_LIB_PATH=paccmann_polymer_amd/libppk_alt.so timeout -k 10 200 python3 -u tools/ws_timing.py $B 2>/dev/null; done; } > $W/ws_timing.txt
fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $W/kt -o run --output-format csv -- python3 $R/tools/small_batch_bench.py 48 3 -1 > $W/kt.log 2>&1
{ echo "# source_hash $H"; echo "# rocprofv3 --kernel-trace --stats -- python3 tools/small_batch_bench.py 48 3 -1   (148 bundle steps, auto schedule)";
  cat $(ls $W/kt/*kernel_stats.csv | head -1); } > $W/small_batch_kernel_stats.csv
