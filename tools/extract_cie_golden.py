#!/usr/bin/env python3
"""tests/golden/cie_d65.json <- the numeric tables of /root/reference/rgb_to_spec/tests/cie_data.rs:10-1907 (CIE 1931 xbar / ybar / zbar
and the normalised D65 curve at 1 nm, 360..830 nm: the reference's OWN second copy of the data its spectrum crate bakes its presets
from).  Data only (decimal literals as written); runs in the build container, the output is committed.  tests/test_oracle.py holds the
baked LUTs of the product (data/presets470.bin, csrc/cie_cmf.inc) against it."""
import json
import os
import re

REF = "/root/reference/rgb_to_spec/tests/cie_data.rs"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = open(REF, encoding="utf-8").read()
    out = {"source": "rgb_to_spec/tests/cie_data.rs:10-1907 (CIE_X, CIE_Y, CIE_Z, D65: 471 samples each, 360..830 nm in 1 nm steps)", "lambda_min": 360, "n": 471}
    for name in ("CIE_X", "CIE_Y", "CIE_Z", "D65"):
        m = re.search(r"const\s+" + name + r"\s*:\s*\[f32;\s*N_CIE_SAMPLES\]\s*=\s*\[(.*?)\];", src, re.S)
        vals = [t.strip() for t in re.sub(r"//[^\n]*", "", m.group(1)).split(",") if t.strip()]
        assert len(vals) == 471, (name, len(vals))
        out[name.lower()] = [float(v) for v in vals]
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "cie_d65.json"), "w"))
    print("wrote tests/golden/cie_d65.json")


if __name__ == "__main__":
    main()
