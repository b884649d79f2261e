# round-6 evidence at HEAD, one gpurun call: the driver's command, rocprofv3 statistics (four engines, steady state; one engine alone),
# PMC passes 3 / 4 / 5 (FETCH_SIZE, WRITE_SIZE, MFMA) of one engine, and the other BASELINE rows with their CPU legs
export ROUND=r06 TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/driver_command.json 2> gpurun_out/r06/driver_command.err; cut -c1-260 gpurun_out/r06/driver_command.json
bash tools/gpu_prof.sh steady 2>&1 | cut -c1-140 | head -8
ISO_WARMUP=8 ISO_STEPS=2 bash tools/gpu_prof.sh iso 2>&1 | cut -c1-140 | head -6
PMC_PASSES="3 4 5" bash tools/gpu_prof.sh pmc 2>&1 | tail -n 6 | cut -c1-200
timeout 900 python bench.py --config 3 > gpurun_out/r06/cfg3_f32.json 2> gpurun_out/r06/cfg3.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg3_f32.json'));print('cfg3',d['value'],d['roofline'].get('frac'),d['roofline'].get('bound'),d['roofline'].get('flop_per_byte'))"
timeout 900 python bench.py --config 4 > gpurun_out/r06/cfg4_f64.json 2> gpurun_out/r06/cfg4.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg4_f64.json'));print('cfg4',d['value'],d['roofline'].get('frac'))"
timeout 900 python bench.py --config 5 > gpurun_out/r06/cfg5_f32.json 2> gpurun_out/r06/cfg5.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg5_f32.json'));print('cfg5',d['value'])"
timeout 900 python bench.py --config 5 --saturated > gpurun_out/r06/cfg5_saturated_f32.json 2> gpurun_out/r06/cfg5s.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg5_saturated_f32.json'));print('cfg5 saturated',d['value'],d['roofline']['frac'])"
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r06/prof_cfg3 -- python3 bench.py --config 3 --no-cpu-baseline > gpurun_out/r06/cfg3_under_rocprof.json 2> gpurun_out/r06/cfg3_rocprof.err
find gpurun_out/r06/prof_cfg3 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06/cfg3_kernel_stats.csv \; ; rm -rf gpurun_out/r06/prof_cfg3; head -n 5 gpurun_out/r06/cfg3_kernel_stats.csv | cut -c1-160
