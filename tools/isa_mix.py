"""Instruction mix of the innermost loop of a kernel, from the device assembly (no GPU needed).

    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only yaqs_amd/csrc/tjm_svd.hip -o /tmp/svd.s
    python tools/isa_mix.py /tmp/svd.s jacobi_cross16x_kernelILi4E

Every VALU instruction of a wave64 occupies the 16-lane SIMD for four cycles on CDNA, an fp64 FMA included (32 flop per clock and
SIMD), so a loop that is bound by VALU issue reaches at most  flop / (2 x VALU instructions)  of the fp64 vector peak: the figure
printed last (for the complex64 build the same ratio against the unpacked fp32 rate).  Used for DESIGN.md section 5 when the PMC counters could not be collected.
"""
import re
import sys


def main(path, needle):
    lines = open(path).read().split("\n")
    start = end = None
    for i, line in enumerate(lines):
        if start is None and line.startswith("_Z") and needle in line.split(":")[0] and ":" in line:
            start = i
        elif start is not None and line.startswith(".Lfunc_end"):
            end = i
            break
    if start is None:
        sys.exit(f"no function matching {needle!r}")
    body = lines[start:end]
    labels = {line.split(":")[0]: i for i, line in enumerate(body) if re.match(r"\.LBB\d+_\d+:", line)}
    loops = []
    for i, line in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", line)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i, m.group(1)))
    lo, hi, name = min(loops, key=lambda t: t[1] - t[0])
    classes = {}
    for line in body[lo: hi + 1]:
        m = re.match(r"\s+([a-z_0-9]+)", line)
        if not m:
            continue
        op = m.group(1)
        if op.startswith(("v_fma_f64", "v_fmac_f64", "v_fma_f32", "v_fmac_f32")):
            k = "fma"
        elif op.startswith("v_pk_fma_f32"):
            k = "packed fma"
        elif op.startswith(("v_pk_mul_f32", "v_pk_add_f32")):
            k = "packed mul / add"
        elif op.startswith(("v_mul_f64", "v_mul_f32")):
            k = "mul"
        elif op.startswith(("v_add_f64", "v_add_f32", "v_sub_f32")):
            k = "add"
        elif "dpp" in op:
            k = "DPP move"
        elif "permlane" in op:
            k = "permlane swap"
        elif op.startswith(("v_readlane", "v_readfirstlane")):
            k = "readlane"
        elif op.startswith("v_cndmask"):
            k = "select"
        elif op.startswith("v_cmp"):
            k = "compare"
        elif op.startswith(("v_rsq", "v_rcp")):
            k = "rsq / rcp"
        elif op.startswith("v_"):
            k = "other VALU"
        elif op.startswith("ds_"):
            k = "LDS"
        elif op.startswith("s_"):
            k = "scalar"
        elif op.startswith(("global", "buffer", "scratch")):
            k = "global memory"
        else:
            k = op
        classes[k] = classes.get(k, 0) + 1
    total = sum(classes.values())
    valu = sum(v for k, v in classes.items() if k not in ("LDS", "scalar", "global memory"))
    flop = (2 * classes.get("fma", 0) + classes.get("mul", 0) + classes.get("add", 0) + 4 * classes.get("packed fma", 0)
            + 2 * classes.get("packed mul / add", 0))
    for k, v in sorted(classes.items(), key=lambda kv: -kv[1]):
        print(f"{k:16s} {v}")
    print(f"innermost loop {name}: {total} instructions, {valu} VALU, {flop} flop per lane and iteration")
    print(f"VALU-issue bound: flop / (2 x VALU instructions) = {flop / (2.0 * valu):.3f} of the unpacked vector FMA rate "
          "(fp64: 78.6 TFLOP/s; fp32: the same unpacked, twice that with v_pk_fma_f32)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
