# round 6, first call: this box's baseline of the headline bench and the complex64 sweep-count A/B of VERDICT r5 item 5
export ROUND=r06
bash tools/gpu_ab.sh "base||" "c64stop03|TJM_MIXED_C64_STOP=0.3|" "c64stop05|TJM_MIXED_C64_STOP=0.5|" "c64cap5|TJM_MIXED_C64_SWEEPS=5|" "c64cap4|TJM_MIXED_C64_SWEEPS=4|"
