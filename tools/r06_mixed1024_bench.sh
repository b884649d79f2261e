# round 6: the fp64 library on bonds of 512 with the mixed-precision split lifted to 1024 rows - full GPU suite, then the chi-saturated
# config-5 layer in complex128 (32 trajectories) with the lift and without (same box)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests -m gpu -q -x -n 4 > gpurun_out/r06/full_suite_c.log 2>&1; tail -n 3 gpurun_out/r06/full_suite_c.log
timeout 1200 python bench.py --config 5 --saturated --dtype complex128 --trajectories 32 --no-cpu-baseline > gpurun_out/r06/cfg5s_f64_mixed1024.json 2> gpurun_out/r06/cfg5s_f64_mixed1024.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg5s_f64_mixed1024.json'));print('cfg5 saturated complex128, mixed split to 1024 rows',d['value'],d['seconds'])"
TJM_MIXED_MAX_DIM=512 timeout 1800 python bench.py --config 5 --saturated --dtype complex128 --trajectories 32 --no-cpu-baseline > gpurun_out/r06/cfg5s_f64_mixed512.json 2> gpurun_out/r06/cfg5s_f64_mixed512.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg5s_f64_mixed512.json'));print('cfg5 saturated complex128, all-fp64 split above 512 rows',d['value'],d['seconds'])"
