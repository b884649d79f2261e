"""fp64 MFMA against fp64 vector FMA on one MI355X: separate pipes or one datapath?  (tools/probes/pipe_probe.hip)

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/pipe_probe.hip -o tools/probes/libpipe_probe.so   # build container
    python tools/probes/pipe_probe.py [blocks=1024] [iters=20000]                                                 # GPU box
"""
import ctypes as C
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libpipe_probe.so"))
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ms = (C.c_float * 5)()
lib.pipe_probe(blocks, iters, ms)
waves = blocks * 4
mfma_flop = waves * iters * 8 * 2048.0
valu_flop = waves * iters * 8 * 16 * 64 * 2.0
out = {
    "blocks": blocks, "iters": iters,
    "valu_alone_ms": ms[0], "valu_TFLOPs": valu_flop / ms[0] / 1e9,
    "mfma_alone_ms": ms[1], "mfma_TFLOPs": mfma_flop / ms[1] / 1e9,
    "interleaved_in_one_wave_ms": ms[2], "two_streams_ms": ms[3], "valu_twice_two_streams_ms": ms[4],
    "sum_of_alone_ms": ms[0] + ms[1], "max_of_alone_ms": max(ms[0], ms[1]),
}
print(json.dumps(out, indent=1))
