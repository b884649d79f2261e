# round 6: complex64 Jacobi tolerance that follows the rows (fp32 inner-product noise): A/B on the chi-saturated circuit layer and on config 3
mkdir -p gpurun_out/r06
for c in 0 1.6 2.5; do
  TJM_JACOBI_TOL_ROWS=$c timeout 900 python bench.py --config 5 --saturated --no-cpu-baseline > gpurun_out/r06/cfg5s_tolrows_$c.json 2> gpurun_out/r06/cfg5s_tolrows_$c.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg5s_tolrows_$c.json'));print('cfg5 saturated, tol rows factor $c:',d['value'],d['roofline']['frac'])"
done
for c in 0 1.6 2.5; do
  TJM_JACOBI_TOL_ROWS=$c timeout 600 python bench.py --config 3 --no-cpu-baseline > gpurun_out/r06/cfg3_tolrows_$c.json 2> gpurun_out/r06/cfg3_tolrows_$c.err; python -c "import json;d=json.load(open('gpurun_out/r06/cfg3_tolrows_$c.json'));print('cfg3, tol rows factor $c:',d['value'],d['mean_Z_site0'])"
done
for c in 0 1.6; do TJM_JACOBI_TOL_ROWS=$c timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -s -k "complex64_library and (512-512 or 320-320)" 2>&1 | grep -E "c64 split|passed|failed" ; done
