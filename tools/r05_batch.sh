export ROUND=r05
mkdir -p gpurun_out/r05
# shard rates
for B in 128 256 512; do timeout 400 python bench.py --batch $B --trajectories $B --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r05/shard_$B.json 2> gpurun_out/r05/shard_$B.err; python -c "import json;d=json.load(open('gpurun_out/r05/shard_$B.json'));print('shard',$B,d['value'])"; done
# pipelined sweeps A/B at the small shard and at the headline
bash tools/gpu_ab.sh "sync128|TJM_SVD_SYNC_EACH=1|--batch 128 --trajectories 128 --steps 4 --warmup 2" "syncall|TJM_SVD_SYNC_EACH=1|--steps 2 --warmup 8" "pipeall||--steps 2 --warmup 8"
