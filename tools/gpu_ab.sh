#!/bin/bash
# A/B runs of the headline bench under diagnostic switches: one JSON line per variant under gpurun_out/r03/ab_<tag>.json
# usage (on the GPU box, from the repository root):  bash tools/gpu_ab.sh "tag1|ENV=1 ...|--extra args" "tag2||..."
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/${ROUND:-r04}
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
    print("   traj/s %.3f  s/step %.2f  svd %.1f TF (%.0f%%)  krylov %.1f TF (%.0f%%)  jacobi %.1f us  sweeps/solve %s" % (
        d["value"], d["ms_per_step"] / 1e3, r["classes"]["svd"]["achieved_TFLOPs"], 100 * r["classes"]["svd"]["share_of_stream_time"],
        r["classes"]["krylov"]["achieved_TFLOPs"], 100 * r["classes"]["krylov"]["share_of_stream_time"],
        r["dominant_kernel"]["avg_launch_us"] or 0.0, r.get("jacobi_sweeps_per_solve")))
except Exception as e:
    print("   FAILED", e)
PY
done
