# round 6, call 7: full GPU suite at HEAD; A/B of the sweeps sequenced inside the library; kernel statistics of the 1024 x 1024 complex64 split
export ROUND=r06 TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests -m gpu -q -x -n 4 > gpurun_out/r06/full_suite_b.log 2>&1; tail -n 3 gpurun_out/r06/full_suite_b.log
timeout 600 python tools/sweep_ab.py 16 16 64 10 > gpurun_out/r06/sweep_ab_L16_chi16_B64.txt 2>&1; tail -n 6 gpurun_out/r06/sweep_ab_L16_chi16_B64.txt
timeout 600 python tools/sweep_ab.py 30 32 256 5 > gpurun_out/r06/sweep_ab_L30_chi32_B256.txt 2>&1; tail -n 6 gpurun_out/r06/sweep_ab_L30_chi32_B256.txt
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r06/q1024 -- python3 tools/svd_bench32.py 32 512 1 > gpurun_out/r06/q1024.log 2>&1
find gpurun_out/r06/q1024 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06/split1024_c64_kernel_stats.csv \; ; rm -rf gpurun_out/r06/q1024
tail -n 1 gpurun_out/r06/q1024.log; head -n 12 gpurun_out/r06/split1024_c64_kernel_stats.csv | cut -d, -f1-5 | cut -c1-160
