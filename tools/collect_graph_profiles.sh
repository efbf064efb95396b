#!/bin/bash
# Runs ON THE GPU BOX: bench_graph.py per case, plain (the roofline JSON lines) and under rocprofv3 FETCH_SIZE /
# WRITE_SIZE passes; writes gpurun_out/profiles/<round>_graph_kernels_roofline.jsonl and ..._graph_kernels_pmc.txt
#   bash tools/collect_graph_profiles.sh r02
set -e -o pipefail
R=$1
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/profiles; mkdir -p $OUT
W=gpurun_out/prof_graph; rm -rf $W; mkdir -p $W
H=$(python3 tools/source_hash.py)
: > $OUT/${R}_graph_kernels_roofline.jsonl
{ echo "# source_hash $H"; echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over: python3 tools/bench_graph.py --only <case>"; } > $OUT/${R}_graph_kernels_pmc.txt
timeout -k 10 200 python3 tools/bench_graph.py --only dense | grep '^{' >> $OUT/${R}_graph_kernels_roofline.jsonl
for c in spmm spmm_random spmm_hubs sparse; do
  timeout -k 10 300 python3 tools/bench_graph.py --only $c | grep '^{' > $W/$c.jsonl
  cat $W/$c.jsonl >> $OUT/${R}_graph_kernels_roofline.jsonl
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $W/${c}_f -o run --output-format csv -- python3 tools/bench_graph.py --only $c > $W/${c}_f.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $W/${c}_w -o run --output-format csv -- python3 tools/bench_graph.py --only $c > $W/${c}_w.log 2>&1
  python3 tools/graph_traffic.py $c $(find $W/${c}_f -name "run_counter_collection.csv") $(find $W/${c}_w -name "run_counter_collection.csv") $W/$c.jsonl >> $OUT/${R}_graph_kernels_pmc.txt
  echo >> $OUT/${R}_graph_kernels_pmc.txt
  echo "case $c done"
done
