# round 6: 1024-row complex64 tile kernel (config 5 saturated A/B), kernel tests at all sizes, config-3 steady fixture, config-3 kernel statistics
export ROUND=r06 TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_hip_kernels.py -x -q -s -k "complex64_library or mixed_split or up_to_1024 or reentrant" > gpurun_out/r06/kernels_x1024.log 2>&1; grep -E "c64 split|passed|failed" gpurun_out/r06/kernels_x1024.log | tail -n 30
timeout 1500 python -m pytest tests/test_hip_fullsize.py -x -q -k "config3" > gpurun_out/r06/fullsize_cfg3.log 2>&1; tail -n 3 gpurun_out/r06/fullsize_cfg3.log
timeout 900 python bench.py --config 5 --saturated --no-cpu-baseline > gpurun_out/r06/cfg5s_x1024.json 2> gpurun_out/r06/cfg5s_x1024.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg5s_x1024.json'));print('cfg5 saturated new',d['value'],d['roofline'].get('frac'))"
TJM_NO_X1024=1 timeout 900 python bench.py --config 5 --saturated --no-cpu-baseline > gpurun_out/r06/cfg5s_old.json 2> gpurun_out/r06/cfg5s_old.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg5s_old.json'));print('cfg5 saturated old',d['value'],d['roofline'].get('frac'))"
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r06/prof_cfg3 -- python3 bench.py --config 3 --no-cpu-baseline > gpurun_out/r06/cfg3_under_rocprof.json 2> gpurun_out/r06/cfg3_rocprof.err
find gpurun_out/r06/prof_cfg3 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06/cfg3_kernel_stats.csv \; ; rm -rf gpurun_out/r06/prof_cfg3; head -n 8 gpurun_out/r06/cfg3_kernel_stats.csv | cut -c1-200
