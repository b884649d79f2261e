# the other BASELINE rows at HEAD (one MI355X): configs 3, 4, 5, 5 saturated, and the shard rates of an 8 / 4 / 2-GPU strong-scaling run
export ROUND=r05
mkdir -p gpurun_out/r05
for B in 128 256 512; do timeout 400 python bench.py --batch $B --trajectories $B --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r05/shard_$B.json 2> gpurun_out/r05/shard_$B.err; python -c "import json;d=json.load(open('gpurun_out/r05/shard_$B.json'));print('shard',$B,d['value'])"; done
timeout 900 python bench.py --config 4 > gpurun_out/r05/cfg4_f64.json 2> gpurun_out/r05/cfg4.err; python -c "import json;d=json.load(open('gpurun_out/r05/cfg4_f64.json'));print('cfg4',d['value'])"
timeout 900 python bench.py --config 3 > gpurun_out/r05/cfg3_f32.json 2> gpurun_out/r05/cfg3.err; python -c "import json;d=json.load(open('gpurun_out/r05/cfg3_f32.json'));print('cfg3',d['value'])"
timeout 900 python bench.py --config 5 --saturated > gpurun_out/r05/cfg5_saturated_f32.json 2> gpurun_out/r05/cfg5s.err; python -c "import json;d=json.load(open('gpurun_out/r05/cfg5_saturated_f32.json'));print('cfg5 saturated',d['value'],d['roofline']['frac'])"
