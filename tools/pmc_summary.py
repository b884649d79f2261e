"""Per-kernel summary of a rocprofv3 --pmc counter_collection.csv (diagnostic tool; output goes under profiles/).

    python tools/pmc_summary.py <counter_collection.csv> <out.csv>

Sums every counter per kernel name; when the MFMA counters are present also derives the MFMA flop count
(SQ_INSTS_VALU_MFMA_MOPS_F64 * 512) and MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * SIMD count): rocprofv3 reports
GRBM_GUI_ACTIVE summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS note), the busy cycles summed over all SIMDs.  With the VALU
counters of the verdict's list (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_INSTS_LDS, SQ_LDS_BANK_CONFLICT, SQ_WAIT_INST_LDS,
SQ_WAVE_CYCLES) it also prints VALU-active and LDS-wait shares of the wave cycles (all three count quad-cycles)."""
import collections
import csv
import sys

SIMDS = 1024  # MI355X: 256 CUs x 4 SIMDs
XCDS = 8      # GRBM_GUI_ACTIVE arrives summed over the XCDs


def main(src, dst):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    first = None
    for r in csv.DictReader(open(src)):
        k = r["Kernel_Name"][:90]
        first = first or r["Counter_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == first:
            calls[k] += 1
    names = sorted({c for v in agg.values() for c in v})
    with open(dst, "w") as out:
        out.write("kernel,dispatches," + ",".join(names) + ",mfma_f64_flop,mfma_util_pct\n")
        order = sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))
        for k, v in order[:16]:
            flop = v.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512
            gui = v.get("GRBM_GUI_ACTIVE", 0.0)
            util = 100.0 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / XCDS * SIMDS) if gui else 0.0
            out.write('"%s",%d,%s,%.6e,%.2f\n' % (k, calls[k], ",".join("%.6e" % v.get(c, 0.0) for c in names), flop, util))
            line = "%s %d mfma flop %.3e MfmaUtil %.1f%%" % (k[:70], calls[k], flop, util)
            wc = v.get("SQ_WAVE_CYCLES", 0.0)
            if wc:
                line += " | VALU active %.1f%% LDS wait %.1f%% of wave cycles" % (100.0 * v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 100.0 * v.get("SQ_WAIT_INST_LDS", 0.0) / wc)
                if v.get("SQ_INSTS_LDS"):
                    line += " | LDS bank-conflict cycles per LDS instruction %.2f" % (v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_INSTS_LDS"])
            print(line)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
