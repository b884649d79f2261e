#!/bin/bash
# rocprofv3 evidence of one headline step: kernel-trace statistics (full batch) and PMC passes of their own (never with a trace
# domain; the program itself after "--").  Output: gpurun_out/r03/prof/  ->  copy the summaries to profiles/.
#   usage (GPU box, repository root):  bash tools/gpu_prof.sh [stats] [steady] [pmc]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/${ROUND:-r05}/prof
mkdir -p "$OUT"
want() { [[ " $* " == *" $WHAT "* ]]; }
for WHAT in "${@:-stats pmc}"; do :; done
ARGS="$*"; [ -z "$ARGS" ] && ARGS="stats pmc"
if [[ " $ARGS " == *" stats "* ]]; then
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/stats_bench.json" 2> "$OUT/stats.err"
  find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
  rm -rf "$OUT/stats"
  head -12 "$OUT/kernel_stats.csv"
fi
if [[ " $ARGS " == *" iso "* ]]; then
  # ONE engine, 256 trajectories: every kernel alone on the device - its duration here is what it costs, not a time slice
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/iso" -- python3 bench.py --steps ${ISO_STEPS:-1} --warmup ${ISO_WARMUP:-1} --no-cpu-baseline --engines 1 --batch 256 --trajectories 256 > "$OUT/iso_bench.json" 2> "$OUT/iso.err"
  find "$OUT/iso" -name "*kernel_stats.csv" -exec cp {} "$OUT/iso_kernel_stats.csv" \;
  rm -rf "$OUT/iso"
  head -12 "$OUT/iso_kernel_stats.csv"
fi
if [[ " $ARGS " == *" steady "* ]]; then
  # the state the driver times: steps 9 - 11 of a run from the Haar state (most dissipations certified), whole trace of the run
  timeout 1500 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/steady" -- python3 bench.py --steps 3 --warmup 8 --no-cpu-baseline > "$OUT/steady_bench.json" 2> "$OUT/steady.err"
  find "$OUT/steady" -name "*kernel_stats.csv" -exec cp {} "$OUT/steady_kernel_stats.csv" \;
  rm -rf "$OUT/steady"
  head -12 "$OUT/steady_kernel_stats.csv"
fi
if [[ " $ARGS " == *" pmc "* ]]; then
  i=0
  for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
             "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" \
             "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    [[ -n "$PMC_PASSES" && " $PMC_PASSES " != *" $i "* ]] && continue  # PMC_PASSES="3 4": only those passes
    # shellcheck disable=SC2086
    timeout 900 rocprofv3 --pmc $pmc -f csv -d "$OUT/pmc$i" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --engines 1 --batch ${PMC_BATCH:-256} --trajectories ${PMC_BATCH:-256} > "$OUT/pmc${i}_bench.json" 2> "$OUT/pmc$i.err"
    csv=$(find "$OUT/pmc$i" -name "*counter_collection.csv" | head -1)
    [ -n "$csv" ] && python3 tools/pmc_summary.py "$csv" "$OUT/pmc${i}_per_kernel.csv" > "$OUT/pmc${i}_summary.txt" 2>&1
    rm -rf "$OUT/pmc$i"
    head -4 "$OUT/pmc${i}_summary.txt"
  done
fi
