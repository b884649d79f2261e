# round 6: the four-column / multi-round complex64 Jacobi at 512 rows (kernel tests, config 3 A/B), the config-4 steady fixture, headline check
export ROUND=r06
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -k "complex64_library or mixed_split or up_to_1024 or reentrant" > gpurun_out/r06/kernels_quad512.log 2>&1; tail -n 3 gpurun_out/r06/kernels_quad512.log
timeout 900 python -m pytest tests/test_hip_fullsize.py -x -q -k "config4" > gpurun_out/r06/fullsize_cfg4.log 2>&1; tail -n 3 gpurun_out/r06/fullsize_cfg4.log
timeout 600 python bench.py --config 3 --no-cpu-baseline > gpurun_out/r06/cfg3_quad512.json 2> gpurun_out/r06/cfg3_quad512.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg3_quad512.json'));print('cfg3 new',d['value'],d['roofline'].get('frac'),d['roofline'].get('bound'))"
TJM_NO_QUAD64_GROUPS=1 timeout 600 python bench.py --config 3 --no-cpu-baseline > gpurun_out/r06/cfg3_q16only.json 2> gpurun_out/r06/cfg3_q16only.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg3_q16only.json'));print('cfg3 four-column, one round per load',d['value'],d['roofline'].get('frac'))"
TJM_QUAD_ONLY_MIXED=1 timeout 600 python bench.py --config 3 --no-cpu-baseline > gpurun_out/r06/cfg3_old.json 2> gpurun_out/r06/cfg3_old.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg3_old.json'));print('cfg3 old (two-column)',d['value'],d['roofline'].get('frac'))"
timeout 600 python bench.py --steps 2 --warmup 8 --no-cpu-baseline > gpurun_out/r06/head_margin.json 2> gpurun_out/r06/head_margin.err; python -c "import json;d=json.load(open('gpurun_out/r06/head_margin.json'));print('headline',d['value'],d['certified_fraction_of_trajectory_steps'])"
