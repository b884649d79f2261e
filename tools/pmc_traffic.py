"""profiles/r01_pmc_traffic.json from the two per-kernel PMC summaries (tools/pmc_summary.py on a FETCH_SIZE pass and on a WRITE_SIZE pass
of `bench.py --steps 1 --warmup 0 --no-cpu-baseline`):
    python tools/pmc_traffic.py <fetch_per_kernel.csv> <write_per_kernel.csv> <L> <chi> <batch> <out.json>
FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md), WRITE_SIZE taken as is; both are in KiB."""
import csv
import json
import sys

KERNEL = "jacobi_cross16x_kernel"


def row(path):
    for r in csv.DictReader(open(path)):
        if KERNEL in r["kernel"]:
            return r
    raise SystemExit(f"{KERNEL} not in {path}")


f, w = row(sys.argv[1]), row(sys.argv[2])
L, chi, batch = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
n = int(f["dispatches"])
fetch_kb = float(f["FETCH_SIZE"]) / n
write_kb = float(w["WRITE_SIZE"]) / int(w["dispatches"])
rec = {"L": L, "chi": chi, "batch": batch, "kernel": KERNEL, "launches": n, "FETCH_SIZE_avg_KB": fetch_kb, "WRITE_SIZE_avg_KB": write_kb,
       "traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
       "note": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 0 --no-cpu-baseline` (B={batch}); "
               f"average over {n} launches; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE taken as is; KB = 1024 B"}
json.dump(rec, open(sys.argv[6], "w"), indent=1)
print(json.dumps(rec))
