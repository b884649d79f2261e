#!/bin/bash
# A/B runs of the headline bench under diagnostic switches: one JSON line per variant under gpurun_out/r03/ab_<tag>.json
# usage (on the GPU box, from the repository root):  bash tools/gpu_ab.sh "tag1|ENV=1 ...|--extra args" "tag2||..."
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/${ROUND:-r05}
mkdir -p "$OUT"
for spec in "$@"; do
  IFS='|' read -r tag envs extra <<< "$spec"
  echo "== $tag ($envs) $extra"
  # shellcheck disable=SC2086
  env $envs timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $extra > "$OUT/ab_$tag.json" 2> "$OUT/ab_$tag.err"
  python - "$OUT/ab_$tag.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    c = r["classes"]
    m = r.get("mixed_split") or {}
    print("   traj/s %.3f  s/step %.2f  svd share %.0f%% (%.1f TF executed)  krylov share %.0f%% (%.1f TF executed)  dominant kernel %.1f us, frac %.3f  c64 sweeps/split %s  fp64 GEMMs/split %s" % (
        d["value"], d["ms_per_step"] / 1e3, 100 * c["svd"]["share_of_stream_time"], c["svd"]["executed_TFLOPs"] or 0.0,
        100 * c["krylov"]["share_of_stream_time"], c["krylov"]["achieved_TFLOPs"] or 0.0, r.get("avg_launch_us") or 0.0, r.get("frac") or 0.0,
        m.get("c64_sweeps_per_split"), m.get("fp64_gemms_per_split")))
except Exception as e:
    print("   FAILED", e)
PY
done
