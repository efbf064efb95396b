#!/bin/bash
# Runs ON THE GPU BOX (gpurun): rocprofv3 kernel stats and PMC passes of the bench command, summarised into
# gpurun_out/profiles/ under the names profiles/ keeps (copy them over afterwards).
#   bash tools/collect_profiles.sh r02 v1 [extra bench args]
# Every profiler pass is its own run of `python3 bench.py` (program directly after `--`), under its own time limit.
set -e -o pipefail
R=$1; V=$2; shift 2
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/profiles; mkdir -p $OUT
H=$(python3 tools/source_hash.py)
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $*"
W=gpurun_out/prof_work; rm -rf $W; mkdir -p $W
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $W/kt -o run --output-format csv -- $CMD > $W/kt.log 2>&1
cp $(find $W/kt -name "run_kernel_stats.csv") $OUT/${R}_bench_kernel_stats_$V.csv
{ echo "# source_hash $H"; echo "# rocprofv3 --kernel-trace of: $CMD  (4 step executions in the trace: 1 warm-up + 2 timed + the profiled extra step; totals over those 4)";
  python3 tools/trace_summary.py $(find $W/kt -name "run_kernel_trace.csv") 1; } > $OUT/${R}_bench_trace_by_shape_$V.txt
CMD1="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $*"
pass() {  # name, counters...
  local n=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" -d $W/$n -o run --output-format csv -- $CMD1 > $W/$n.log 2>&1
  python3 tools/pmc_summary.py $(find $W/$n -name "run_counter_collection.csv")
}
{ echo "# source_hash $H"
  echo "# rocprofv3 --pmc passes over: $CMD1  (3 training steps incl. the profiled one)"
  echo "# GRBM_GUI_ACTIVE is summed over the 8 XCDs; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)"
  pass sq GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES
  echo
  echo "# FETCH_SIZE / WRITE_SIZE in KiB as reported (gfx950: FETCH_SIZE counts half of wide streaming reads, double it)"
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  echo
  echo "# L2: hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)"
  pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
} > $OUT/${R}_bench_pmc_$V.txt
echo "profiles written for source_hash $H:"; ls -la $OUT
