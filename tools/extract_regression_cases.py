#!/usr/bin/env python3
"""Writes tests/golden/regression_cases.json: the 42 golden-image regression cases of the reference as DATA — (scene, renderer, sampler,
spp, 200x150, reference file name, RMSE threshold) from renderer/tests/regression_test.rs:109-659 and the git-LFS identity (sha256 oid,
size) of each reference PNG from the pointer stubs under test_references/.  Run in the build container (needs /root/reference);
tools/run_reference_regressions.py and the tests read only the JSON."""
import json, os, re, sys
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(REF, "renderer", "tests", "regression_test.rs")).read()
calls = re.findall(r'run_render_and_compare\(\s*(\d+),\s*"(\w+)",\s*"(\w+)",\s*(\d+),\s*"([^"]+)",\s*"([^"]+)",\s*([\d.]+),?\s*\)', src)
cases = []
for scene, renderer, sampler, spp, out, ref, thr in calls:
    stub = open(os.path.join(REF, ref)).read()
    m = re.search(r"oid sha256:([0-9a-f]{64})\s+size (\d+)", stub)
    cases.append({"scene": int(scene), "renderer": renderer, "sampler": sampler, "spp": int(spp), "width": 200, "height": 150,
                  "output": out, "reference": ref, "max_rmse": float(thr), "sha256": m.group(1), "size": int(m.group(2))})
json.dump({"source": "renderer/tests/regression_test.rs:109-659 + test_references/*.png (git-LFS pointer stubs)", "cases": cases},
          open(os.path.join(ROOT, "tests", "golden", "regression_cases.json"), "w"), indent=1)
print(len(cases), "cases")
