#!/bin/bash
# Second GPU call of the next round, in a call of its own (round 2 lost two boxes to the FIRST version of this test - its checker
# contracted chi^4 transfer tensors on the host; fixed, and the gate / shift half passes on tests/hipsim):
#   gpurun --timeout 1500 -- 'bash tools/gpu_session_chi512.sh'
# If both tests pass, raise MAX_CHI in yaqs_amd/tjm.py to 512 (bonds above 256 at the Simulator level, row J2 of DESIGN section 7).
set -u
OUT=gpurun_out/chi512
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TJM_TEST_CHI512_ENGINE=1
# host memory of the oracle side: 0.6 GB, 70 s (no ulimit -v here: the ROCm runtime reserves far more address space than it uses)
timeout 600 python -m pytest tests/test_hip_round2.py -k bonds_up_to_512_gate -x -q > "$OUT/gate_and_shifts.log" 2>&1
echo "gate / shifts: exit $?" | tee "$OUT/status.txt"
timeout 700 python -m pytest tests/test_hip_round2.py -k bonds_up_to_512_two_site -x -q > "$OUT/two_site_sweep.log" 2>&1
echo "two-site sweep: exit $?" | tee -a "$OUT/status.txt"
