# round 6: do more HIP hardware queues let more engines overlap?  (the runtime maps streams onto GPU_MAX_HW_QUEUES = 4 queues by default)
mkdir -p gpurun_out/r06
run() { # tag, env, args
  env $2 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline $3 > gpurun_out/r06/hwq_$1.json 2> gpurun_out/r06/hwq_$1.err; python -c "import json;d=json.load(open('gpurun_out/r06/hwq_$1.json'));print('$1', d['value'])"
}
run s128_e4_q4 "GPU_MAX_HW_QUEUES=4" "--batch 128 --trajectories 128 --engines 4"
run s128_e8_q4 "GPU_MAX_HW_QUEUES=4" "--batch 128 --trajectories 128 --engines 8"
run s128_e8_q8 "GPU_MAX_HW_QUEUES=8" "--batch 128 --trajectories 128 --engines 8"
run s128_e4_q8 "GPU_MAX_HW_QUEUES=8" "--batch 128 --trajectories 128 --engines 4"
run s128_e16_q16 "GPU_MAX_HW_QUEUES=16" "--batch 128 --trajectories 128 --engines 16"
run b1024_e4_q4 "GPU_MAX_HW_QUEUES=4" "--engines 4"
run b1024_e8_q8 "GPU_MAX_HW_QUEUES=8" "--engines 8"
run b1024_e6_q8 "GPU_MAX_HW_QUEUES=8" "--engines 6"
