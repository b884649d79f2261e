# round 6 experiment: the mixed-precision split for the fp64 library's 1024 x 1024 splits (bonds up to 512), env-gated
mkdir -p gpurun_out/r06
for v in 512 1024; do
  echo "== TJM_MIXED_MAX_DIM=$v"
  TJM_MIXED_MAX_DIM=$v TJM_DEBUG_SVD=1 timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -k "up_to_1024" --durations=5 > gpurun_out/r06/mixed1024_kernels_$v.log 2>&1
  grep -E "passed|failed|svd-mixed|^[0-9.]+s (call|setup)" gpurun_out/r06/mixed1024_kernels_$v.log | sort | uniq -c | sort -rn | head -n 12
  TJM_MIXED_MAX_DIM=$v timeout 1200 python -m pytest tests/test_hip_round2.py -x -q -k "bonds_up_to_512 or reach_512" --durations=5 > gpurun_out/r06/mixed1024_engine_$v.log 2>&1
  grep -E "passed|failed|^[0-9.]+s (call|setup)" gpurun_out/r06/mixed1024_engine_$v.log | head -n 8
done
