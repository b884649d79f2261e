# round 6: PMC counters of the complex64 split's kernels at 512 x 512 and 256 x 256 (one stream, 32 / 128 matrices): VALU activity,
# waits, LDS.  Counter passes of their own (never with a trace domain), the program itself after "--".
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  i=$((i + 1))
  for chi in 256 128; do
    # shellcheck disable=SC2086
    timeout 300 rocprofv3 --pmc $pmc -f csv -d gpurun_out/r06/qpmc${i}_$chi -- python3 tools/svd_bench32.py 32 $chi 1 > gpurun_out/r06/qpmc${i}_$chi.log 2>&1
    csv=$(find gpurun_out/r06/qpmc${i}_$chi -name "*counter_collection.csv" | head -1)
    [ -n "$csv" ] && python3 tools/pmc_summary.py "$csv" gpurun_out/r06/q64_pmc${i}_chi${chi}_per_kernel.csv > gpurun_out/r06/q64_pmc${i}_chi${chi}_summary.txt 2>&1
    rm -rf gpurun_out/r06/qpmc${i}_$chi
    grep -E "quad64|qr_block_apply_multi|qr_panel" gpurun_out/r06/q64_pmc${i}_chi${chi}_summary.txt | cut -c1-260
  done
done
