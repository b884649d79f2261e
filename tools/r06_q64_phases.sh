# round 6: where does a launch of the 512-row four-block kernel spend its time?  One stream, nothing else on the device.
export ROUND=r06 TMPDIR=/tmp
mkdir -p gpurun_out/r06
for tag in full load; do
  for B in 32 128; do
    if [ $tag = load ]; then export TJM_Q64_ABLATE=1; else unset TJM_Q64_ABLATE; fi
    timeout 300 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r06/q64_${tag}_$B -- python3 tools/svd_bench32.py $B 256 2 > gpurun_out/r06/q64_${tag}_$B.log 2>&1
    find gpurun_out/r06/q64_${tag}_$B -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06/q64_${tag}_${B}_kernel_stats.csv \; ; rm -rf gpurun_out/r06/q64_${tag}_$B
    tail -n 1 gpurun_out/r06/q64_${tag}_$B.log; grep -E "quad64|cross16" gpurun_out/r06/q64_${tag}_${B}_kernel_stats.csv | cut -d, -f1-5 | cut -c1-200
  done
done
unset TJM_Q64_ABLATE
for v in "TJM_NO_QUAD64_GROUPS=1" "TJM_QUAD_ONLY_MIXED=1"; do
  env $v timeout 300 python3 tools/svd_bench32.py 128 256 2 2>&1 | tail -n 1 | sed "s/^/$v: /"
done
timeout 300 python3 tools/svd_bench32.py 32 512 1 2>&1 | tail -n 1
TJM_NO_X1024=1 timeout 300 python3 tools/svd_bench32.py 32 512 1 2>&1 | tail -n 1 | sed "s/^/TJM_NO_X1024=1: /"
