#!/bin/bash
# Same-box comparison of several library builds: every variant once per round, rounds in forward then reverse order
# (box-to-box spread ~3 %; order effects cancel).  Runs ON THE GPU BOX:
#   bash tools/multi_bench.sh "base=paccmann_polymer_amd/libppk.so sc1=paccmann_polymer_amd/libppk_sc1.so" [steps] [extra bench args]
VARS=$1; STEPS=${2:-10}; shift 2 || true
fwd=($VARS); rev=(); for ((i=${#fwd[@]}-1; i>=0; i--)); do rev+=("${fwd[i]}"); done
for v in "${fwd[@]}" "${rev[@]}"; do
  name=${v%%=*}; lib=${v#*=}
  echo -n "[$name] "
  PPK_LIB_PATH=$lib timeout -k 10 300 python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 \
    | grep -o "ms_per_step[^,]*\|\"k_seq_bwd\": [0-9.]*\|\"k_seq_fwd\": [0-9.]*\|\"k_tn128\": [0-9.]*" | tr "\n" " " || exit 1
  echo
done
