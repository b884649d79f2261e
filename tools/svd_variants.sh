#!/bin/bash
# kernel-level A/B of the batched two-site split under environment switches: tools/svd_bench.py under rocprofv3 --kernel-trace --stats
#   usage (GPU box, repository root):  bash tools/svd_variants.sh "tag|ENV=1 ..." ...
cd "$(dirname "$0")/.." || exit 1; export TMPDIR=/tmp; mkdir -p gpurun_out/r03/svdv
for spec in "$@"; do
  IFS='|' read -r tag envs <<< "$spec"
  # shellcheck disable=SC2086
  env $envs timeout 300 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r03/svdv/$tag -- python3 tools/svd_bench.py 1024 128 1 > gpurun_out/r03/svdv/$tag.txt 2>&1
  f=$(find gpurun_out/r03/svdv/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag ($envs)"; grep "ms per batched" gpurun_out/r03/svdv/$tag.txt; grep -i "jacobi_" "$f" | cut -c1-220 | head -6
  cp "$f" gpurun_out/r03/svdv/${tag}_kernel_stats.csv 2>/dev/null; rm -rf gpurun_out/r03/svdv/$tag
done
