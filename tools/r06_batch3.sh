# round 6, call 4: full GPU suite at HEAD; shard rates (launch-count reductions); headline steady window; config 3 engines A/B; config 5 saturated
export ROUND=r06 TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests -m gpu -q -x -n 4 > gpurun_out/r06/full_suite_a.log 2>&1; tail -n 3 gpurun_out/r06/full_suite_a.log
for B in 128 256 512; do timeout 400 python bench.py --batch $B --trajectories $B --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r06/shard_$B.json 2> gpurun_out/r06/shard_$B.err; python -c "import json;d=json.load(open('gpurun_out/r06/shard_$B.json'));print('shard',$B,d['value'])"; done
timeout 600 python bench.py --steps 2 --warmup 8 --no-cpu-baseline > gpurun_out/r06/head_b.json 2> gpurun_out/r06/head_b.err; python -c "import json;d=json.load(open('gpurun_out/r06/head_b.json'));print('headline steps 9-10',d['value'],d['config']['certified_fraction_of_trajectory_steps'])"
timeout 600 python bench.py --config 3 --no-cpu-baseline > gpurun_out/r06/cfg3_e4.json 2> gpurun_out/r06/cfg3_e4.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg3_e4.json'));print('cfg3 4 engines',d['value'],d['roofline'].get('frac'))"
timeout 600 python bench.py --config 3 --no-cpu-baseline --engines 2 > gpurun_out/r06/cfg3_e2.json 2> gpurun_out/r06/cfg3_e2.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg3_e2.json'));print('cfg3 2 engines',d['value'],d['roofline'].get('frac'))"
timeout 600 python bench.py --config 3 --no-cpu-baseline --engines 8 > gpurun_out/r06/cfg3_e8.json 2> gpurun_out/r06/cfg3_e8.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg3_e8.json'));print('cfg3 8 engines',d['value'],d['roofline'].get('frac'))"
timeout 900 python bench.py --config 5 --saturated --no-cpu-baseline > gpurun_out/r06/cfg5s_b.json 2> gpurun_out/r06/cfg5s_b.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg5s_b.json'));print('cfg5 saturated',d['value'],d['roofline'].get('frac'))"
