#!/usr/bin/env python3
"""Bake the reference's preset spectra into 470-entry f32 LUTs (data, not code).

Reads the *numeric tables* of /root/reference/spectrum/src/presets.rs as text and
evaluates them exactly the way the reference does at start-up:

  * PiecewiseLinearSpectrum::value  (spectrum/src/spectrum/piecewise_linear_spectrum.rs:66-80)
  * DenselySampledSpectrum::from    (spectrum/src/spectrum/densely_sampled_spectrum.rs:40-53)
    -> values[i] = spec.value(360 + i), i in 0..470
  * illuminant normalisation by sum_i s(360+i) * ybar(360+i)
    (piecewise_linear_spectrum.rs:47-60, spectrum.rs:67-78)

The reference's CIE_LAMBDA table contains a few typo'd entries (e.g. "36.01" for 361);
because the bake follows the reference's lookup loop literally, the resulting LUTs carry
the same (tiny) one-entry shifts as the reference's runtime LUTs.

Outputs (committed, little-endian f32):
  toy-cpu-pathtracing_amd/data/presets470.bin    N x 470 f32
  toy-cpu-pathtracing_amd/data/presets470.json   {"names": [...], "n": 470}
  tests/golden/sobol_matrices_dim01.json         first 2x52 words of the Sobol table

This script only runs in the build container (the reference tree is not on the GPU box);
its outputs are committed.
"""
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


def parse_arrays(path):
    src = open(path, encoding="utf-8").read()
    out = {}
    for m in re.finditer(r"const\s+([A-Z0-9_]+)\s*:\s*[^=]*=\s*&?\[(.*?)\];", src, re.S):
        name, body = m.group(1), m.group(2)
        body = re.sub(r"//[^\n]*", "", body)
        toks = [t.strip().replace("_", "") for t in body.split(",") if t.strip()]
        try:
            out[name] = np.array([float(t) for t in toks], dtype=np.float64).astype(F)
        except ValueError:
            pass
    return out


def pw_value(lams, vals, lam):
    """piecewise_linear_spectrum.rs:66-80, f32 arithmetic."""
    lam = F(lam)
    n = len(lams)
    if n == 0:
        return F(0)
    if lam < lams[0] or lam > lams[n - 1]:
        return F(0)
    i = 0
    while i < n - 1 and lams[i + 1] < lam:
        i += 1
    t = F(F(lam - lams[i]) / F(lams[i + 1] - lams[i]))
    return F(F(vals[i] * F(F(1) - t)) + F(vals[i + 1] * t))


def dense(lams, vals):
    return np.array([pw_value(lams, vals, F(360.0) + F(i)) for i in range(470)], dtype=F)


def main():
    arrs = parse_arrays(os.path.join(REF, "spectrum/src/presets.rs"))
    lam = arrs["CIE_LAMBDA"]
    bad = [(i, float(lam[i])) for i in range(len(lam)) if lam[i] != F(360 + i)]
    print("CIE_LAMBDA entries that differ from 360+i:", bad)
    luts = {}
    for k in ("X", "Y", "Z"):
        luts["cie_" + k.lower()] = dense(lam, arrs["CIE_" + k])
    ybar = luts["cie_y"]

    def interleaved(name, normalized):
        a = arrs[name]
        l, v = a[0::2].copy(), a[1::2].copy()
        d = dense(l, v)
        if normalized:
            s = F(0)
            for i in range(470):           # inner_product, f32 running sum
                s = F(s + F(d[i] * ybar[i]))
            if s == 0:
                return np.zeros(470, F)
            d = np.array([F(x / s) for x in d], dtype=F)
        return d

    for name in arrs:
        if name.startswith("CIE_ILLUM_") or name == "ACES_ILLUM_D60":
            luts[name.lower()] = interleaved(name, True)
        elif name.endswith("_ETA") or name.endswith("_K"):
            luts[name.lower()] = interleaved(name, False)

    names = sorted(luts)
    data = np.stack([luts[n] for n in names]).astype("<f4")
    ddir = os.path.join(ROOT, "toy-cpu-pathtracing_amd", "data")
    os.makedirs(ddir, exist_ok=True)
    data.tofile(os.path.join(ddir, "presets470.bin"))
    json.dump({"names": names, "n": 470}, open(os.path.join(ddir, "presets470.json"), "w"), indent=1)
    print("baked", len(names), "LUTs:", names)

    # Sobol: first 2 dimensions x 52 columns (the only words the sampler reads,
    # renderer/src/sampler/z_sobol_sampler.rs:158-177,208,221,225)
    src = open(os.path.join(REF, "renderer/src/sampler/sobol_matrices.rs")).read()
    body = src[src.index("= [") + 3:]
    words = re.findall(r"0x([0-9a-fA-F]{8})", body[:4000])[:104]
    gdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gdir, exist_ok=True)
    json.dump({"source": "renderer/src/sampler/sobol_matrices.rs:7 (first 104 words)",
               "words": [int(w, 16) for w in words]},
              open(os.path.join(gdir, "sobol_matrices_dim01.json"), "w"))
    print("sobol words", len(words))


if __name__ == "__main__":
    sys.exit(main())
