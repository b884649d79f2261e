# round-5 evidence at HEAD, one gpurun call: full GPU suite, the driver's command, rocprofv3 statistics (four engines, steady state; one
# engine alone), PMC passes 3 / 4 (FETCH_SIZE, WRITE_SIZE) for the HBM bytes per launch of the dominant kernel
export ROUND=r05
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -q -x -n 4 > gpurun_out/r05/full_suite_head.log 2>&1; tail -3 gpurun_out/r05/full_suite_head.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/driver_command.json 2> gpurun_out/r05/driver_command.err; cut -c1-200 gpurun_out/r05/driver_command.json
bash tools/gpu_prof.sh steady 2>&1 | cut -c1-120 | head -6
ISO_WARMUP=8 ISO_STEPS=2 bash tools/gpu_prof.sh iso 2>&1 | cut -c1-120 | head -4
PMC_PASSES="3 4" bash tools/gpu_prof.sh pmc 2>&1 | tail -4
