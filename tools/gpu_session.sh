#!/bin/bash
# One GPU call that collects the evidence round 2 could not (the boxes were lost before): run from the repository root on the GPU box,
#   gpurun --timeout 3000 -- 'bash tools/gpu_session.sh'
# Everything is written under gpurun_out/session/ (merged back by gpurun); copy what is to be judged into profiles/ afterwards.
# Every step has its own timeout; a failing step does not stop the next one.  The chi = 512 ENGINE test is NOT part of this script
# (it is the test during which two boxes went down in round 2; its checker is fixed, run it in a call of its own:
#   gpurun --timeout 1500 -- 'bash tools/gpu_session_chi512.sh').
set -u
OUT=gpurun_out/session
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp

step() {  # step <name> <seconds> <command...>
  local name=$1 limit=$2
  shift 2
  echo "== $name" | tee -a "$OUT/steps.log"
  local t0=$SECONDS
  timeout "$limit" "$@" > "$OUT/$name.log" 2>&1
  echo "   exit $? after $((SECONDS - t0)) s" | tee -a "$OUT/steps.log"
}

# 1. parity: the whole -m gpu suite (new since the last full run: everything written after the box losses of round 2)
step pytest_gpu 1500 python -m pytest tests -m gpu -q --durations=25
# 2. the headline bench, fp64 (reference precision) and the first timing of the complex64 build
step bench_f64 600 python bench.py --steps 3 --warmup 1
step bench_f32 400 python bench.py --steps 3 --warmup 1 --dtype complex64 --no-cpu-baseline
# 3. A/B of the switches that have never been measured
step bench_krylov_sync_each 300 env TJM_KRYLOV_SYNC_EACH=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline
step bench_svd_chunk192 300 env TJM_SVD_CHUNK=192 python bench.py --steps 2 --warmup 1 --no-cpu-baseline
step bench_engines2_locks 300 env TJM_PHASE_LOCKS=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --engines 2
step physics_probe 600 python tests/probes/physics_probe.py 1024 --steps 40 --check 3
step physics_probe_sync_each 600 env TJM_KRYLOV_SYNC_EACH=1 python tests/probes/physics_probe.py 1024 --steps 40 --check 0
# 4. profiles of one headline step (the program itself after "--": no env / bash -c hops under rocprofv3)
step rocprof_stats 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_stats" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo "$pmc" | tr ' ' '_' | cut -c1-40)
  # counters in runs of their own, never together with a trace domain (gpurun refuses that combination)
  step "pmc_$tag" 900 rocprofv3 --pmc $pmc -d "$OUT/pmc_$tag" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --batch 256 --trajectories 256
  # gpurun merges at most 64 MiB back: keep the per-kernel sums, drop the per-dispatch rows
  csv=$(find "$OUT/pmc_$tag" -name "*counter_collection.csv" | head -1)
  [ -n "$csv" ] && python3 tools/pmc_summary.py "$csv" "$OUT/pmc_${tag}_per_kernel.csv" > "$OUT/pmc_${tag}_summary.txt" 2>&1
  rm -rf "$OUT/pmc_$tag"
done
# of the kernel trace keep the statistics only
find "$OUT/prof_stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/prof_stats"
ls -R "$OUT" | head -100 > "$OUT/files.txt"
echo "done" | tee -a "$OUT/steps.log"
