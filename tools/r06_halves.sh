# round 6: the four-block Jacobi kernel for 512 rows with the rows of a column split over two wavefronts (four wavefronts per SIMD)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -k "complex64_library or mixed_split" 2>&1 | tail -n 2
timeout 900 python -m pytest tests/test_hip_fullsize.py -x -q -k "config3" 2>&1 | tail -n 2
for v in halves whole; do
  if [ $v = whole ]; then export TJM_NO_QUAD64_HALVES=1; else unset TJM_NO_QUAD64_HALVES; fi
  timeout 300 python3 tools/svd_bench32.py 128 256 2 2>&1 | tail -n 1 | sed "s/^/$v: /"
  timeout 300 python3 tools/svd_bench32.py 32 256 2 2>&1 | tail -n 1 | sed "s/^/$v: /"
  timeout 600 python bench.py --config 3 --no-cpu-baseline > gpurun_out/r06/cfg3_$v.json 2> gpurun_out/r06/cfg3_$v.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg3_$v.json'));print('cfg3 $v',d['value'],d['roofline'].get('frac'))"
done
unset TJM_NO_QUAD64_HALVES
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r06/q64h -- python3 tools/svd_bench32.py 32 256 2 > gpurun_out/r06/q64h_32.log 2>&1
find gpurun_out/r06/q64h -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06/q64h_32_kernel_stats.csv \; ; rm -rf gpurun_out/r06/q64h; grep -E "quad64" gpurun_out/r06/q64h_32_kernel_stats.csv | cut -d, -f1-5 | cut -c1-200
