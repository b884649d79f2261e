#!/bin/bash
# PMC passes over tools/gemm_bench.py at one K (both instruction forms): MFMA busy cycles, wave stalls.  Output gpurun_out/r05g/pmc_*.txt
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05g
mkdir -p "$OUT"
K=${1:-512}
for mode in old new; do
  if [ $mode = new ]; then export TJM_GEMM_4X4=1; else unset TJM_GEMM_4X4; fi
  i=0
  for pmc in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
    i=$((i + 1))
    # shellcheck disable=SC2086
    timeout 300 rocprofv3 --pmc $pmc -f csv -d "$OUT/pmc_$mode$i" -- python3 tools/gemm_bench.py 256 10 K=$K > "$OUT/pmc_${mode}${i}_bench.txt" 2> "$OUT/pmc_$mode$i.err"
    csv=$(find "$OUT/pmc_$mode$i" -name "*counter_collection.csv" | head -1)
    [ -n "$csv" ] && python3 tools/pmc_summary.py "$csv" "$OUT/pmc_${mode}${i}_per_kernel.csv" > "$OUT/pmc_${mode}${i}_summary.txt" 2>&1
    rm -rf "$OUT/pmc_$mode$i"
    echo "== $mode pass $i K=$K"; grep -v amdgpu.ids "$OUT/pmc_${mode}${i}_bench.txt"; grep -i gemm "$OUT/pmc_${mode}${i}_per_kernel.csv"
  done
done
