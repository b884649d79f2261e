"""Golden vectors held by the reference's own tests, as data: the three two-site-correlator series of
/root/reference/tests/test_simulator.py:858-1188 (closed 4-site Ising chain, <XX>, <YY>, <ZZ> on the left, centre and right pair at
21 time points) -> tests/golden/reference_two_site_correlators.json.  Run in the build container (the reference is not on the GPU box)."""
import json
import re

src = open("/root/reference/tests/test_simulator.py").read()
pairs = {"test_two_site_correlator_left_boundary": (0, 1), "test_two_site_correlator_center": (2, 3), "test_two_site_correlator_right_boundary": (2, 3)}
out = {}
for fn, sites in pairs.items():
    body = re.search(r"def " + fn + r"\(\).*?(?=\ndef |\Z)", src, re.S).group(0)
    rec = {"sites": list(sites), "L": int(re.search(r"L = (\d+)", body).group(1)), "J": float(re.search(r"J = ([0-9.]+)", body).group(1)),
           "g": float(re.search(r"g = ([0-9.]+)", body).group(1)), "elapsed_time": float(re.search(r"elapsed_time=([0-9.]+)", body).group(1)),
           "dt": float(re.search(r"dt=([0-9.]+)", body).group(1)), "max_bond_dim": int(re.search(r"max_bond_dim=(\d+)", body).group(1))}
    for name in ("xx", "yy", "zz"):
        arr = re.search(r"expected_" + name + r" = np\.array\(\[(.*?)\]\)", body, re.S).group(1)
        rec[name] = [float(x) for x in re.findall(r"[-+]?\d+\.\d+(?:e[-+]?\d+)?", arr)]
        assert len(rec[name]) == 21
    out[fn] = rec
json.dump(out, open("tests/golden/reference_two_site_correlators.json", "w"), indent=1)
print({k: (v["sites"], v["L"], v["g"]) for k, v in out.items()})
