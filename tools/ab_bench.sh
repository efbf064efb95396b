#!/bin/bash
# Same-box A/B of bench.py under two environments, in ABBA order (box-to-box spread is ~3 %, larger than most of the
# effects worth measuring; thermal drift inside one call is cancelled by the order).  Runs ON THE GPU BOX:
#   bash tools/ab_bench.sh "PPK_LIB_PATH=paccmann_polymer_amd/libppk_base.so" "PPK_SEQ_XT=8" [steps] [extra bench args]
A=$1; B=$2; STEPS=${3:-20}; shift 3 || true
for cfg in "$A" "$B" "$B" "$A" "$A" "$B" "$B" "$A"; do
  echo -n "[$cfg] "
  env $cfg timeout -k 10 300 python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 \
    | grep -o "ms_per_step[^,]*\|\"k_seq_bwd\": [0-9.]*\|\"k_seq_fwd\": [0-9.]*\|\"k_tn128\": [0-9.]*" | tr "\n" " "
  echo
done
