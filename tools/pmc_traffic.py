"""profiles/r01_pmc_traffic.json from the two per-kernel PMC summaries (tools/pmc_summary.py on a FETCH_SIZE pass and on a WRITE_SIZE pass
of `bench.py --steps 1 --warmup 0 --no-cpu-baseline`):
    python tools/pmc_traffic.py <fetch_per_kernel.csv> <write_per_kernel.csv> <L> <chi> <batch> <out.json> [kernel substring] [steps]
(kernel substring: default "tjm::" = the fp64 tile kernels jacobi_cross16x_kernel; "tjm32::" selects the complex64 instances of the
mixed-precision split - jacobi_cross16q_kernel, four columns per wavefront, since round 4)
FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md), WRITE_SIZE taken as is; both are in KiB."""
import csv
import json
import sys

import re

KERNEL = sys.argv[9] if len(sys.argv) > 9 else "jacobi_cross16"  # ..x_kernel / ..q_kernel; "jacobi_quad64" = three rounds per load (round 5)
TAG = sys.argv[7] if len(sys.argv) > 7 else "tjm::"
STEPS = int(sys.argv[8]) if len(sys.argv) > 8 else 1


def row(path):
    """all launches of the tile kernel (both template instances) in the chosen namespace"""
    acc = None
    for r in csv.DictReader(open(path)):
        if KERNEL in r["kernel"] and r["kernel"].lstrip("void ").startswith(TAG):
            if acc is None:
                acc = dict(r)
                for k in acc:
                    if k not in ("kernel",):
                        acc[k] = float(acc[k])
            else:
                for k in r:
                    if k not in ("kernel",):
                        acc[k] += float(r[k])
    if acc is None:
        raise SystemExit(f"{TAG}...{KERNEL} not in {path}")
    return acc


def names(path):
    return sorted({re.search(KERNEL + r"\w*", r["kernel"]).group(0) for r in csv.DictReader(open(path))
                   if KERNEL in r["kernel"] and r["kernel"].lstrip("void ").startswith(TAG)})


f, w = row(sys.argv[1]), row(sys.argv[2])
L, chi, batch = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
n = int(f["dispatches"])
fetch_kb = float(f["FETCH_SIZE"]) / n
write_kb = float(w["WRITE_SIZE"]) / int(w["dispatches"])
rec = {"L": L, "chi": chi, "batch": batch, "kernel": TAG + "(anonymous namespace)::" + "/".join(names(sys.argv[1])), "steps": STEPS, "launches": n, "FETCH_SIZE_avg_KB": fetch_kb, "WRITE_SIZE_avg_KB": write_kb,
       "traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
       "note": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 0 --no-cpu-baseline` (B={batch}); "
               f"average over {n} launches; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE taken as is; KB = 1024 B"}
json.dump(rec, open(sys.argv[6], "w"), indent=1)
print(json.dumps(rec))
