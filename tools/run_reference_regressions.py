#!/usr/bin/env python3
"""The reference's 42 golden-image regressions (renderer/tests/regression_test.rs) against the mi355pt CLI.

For every case of tests/golden/regression_cases.json: find the reference PNG under --references, check it against the sha256 oid of its
git-LFS pointer stub (the stubs are all this repository has ever seen: the objects are not in the reference checkout), render the case with
`toy-cpu-pathtracing_amd/host/mi355pt` using the reference's own CLI invocation (regression_test.rs:53-63: --scene --renderer --sampler --spp
--width 200 --height 150 --output) and compare with the reference's metric (regression_test.rs:6-40: RMSE over the RGB bytes after the sRGB
EOTF) against the case's threshold.  A holder of the LFS objects runs

    tools/run_reference_regressions.py --references <checkout>/test_references --assets <checkout>/renderer/assets

on a GPU box; that run is what turns "parity unpinned" into "pinned".  With the stubs only (or no directory) it reports
"unpinned: N/42 references absent" and exits 0.  Exit 1: a supplied reference failed its threshold or its checksum.
--assets: directory with the reference's real OBJ / PNG / EXR assets (default: the synthetic stand-ins, exported on the fly — RMSE against
the real golden images is then NOT meaningful and the report says so)."""
import argparse, hashlib, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def srgb_to_linear(s):
    import numpy as np
    return np.where(s <= 0.04045, s / 12.92, ((s + 0.055) / 1.055) ** 2.4)


def rmse_linear_u8(a, b):            # regression_test.rs:6-40
    import numpy as np
    d = srgb_to_linear(a.astype(np.float64) / 255.0) - srgb_to_linear(b.astype(np.float64) / 255.0)
    return float(np.sqrt(np.mean(d * d)))


def reference_state(path, case, check=True):
    """'absent' (missing or an LFS pointer stub), 'checksum' (present, wrong bytes) or 'ok'"""
    if not path or not os.path.isfile(path):
        return "absent"
    data = open(path, "rb").read()
    if data.startswith(b"version https://git-lfs"):
        return "absent"
    return "ok" if not check or (hashlib.sha256(data).hexdigest() == case["sha256"] and len(data) == case["size"]) else "checksum"


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--references", default=None, help="directory holding reference_*.png (the reference's test_references/)")
    ap.add_argument("--assets", default=None, help="directory with the reference's real assets; default: synthetic stand-ins")
    ap.add_argument("--cli", default=os.path.join(ROOT, "toy-cpu-pathtracing_amd", "host", "mi355pt"))
    ap.add_argument("--only", default=None, help="substring filter on the reference file name")
    ap.add_argument("--json", default=None, help="write the per-case report here")
    ap.add_argument("--rehearsal", action="store_true", help="skip the sha256 check (plumbing test with self-rendered 'references'); never reports 'pinned'")
    a = ap.parse_args(argv)
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "regression_cases.json")))["cases"]
    if a.only:
        cases = [c for c in cases if a.only in c["reference"]]
    report, absent, failed = [], 0, 0
    todo = []
    for c in cases:
        path = os.path.join(a.references, os.path.basename(c["reference"])) if a.references else None
        st = reference_state(path, c, check=not a.rehearsal)
        if st == "absent":
            absent += 1
        elif st == "checksum":
            failed += 1
            print(f"CHECKSUM  {c['reference']}: not the object the reference's LFS pointer names (sha256 {c['sha256'][:12]}…, {c['size']} B)")
        else:
            todo.append((c, path))
        report.append({**c, "state": st})
    if todo:
        import numpy as np
        from PIL import Image
        assets = a.assets
        tmp = tempfile.mkdtemp(prefix="mi355pt_reg_")
        if assets is None:
            assets = os.path.join(tmp, "assets")
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "export_assets.py"), assets])
            print("NOTE: rendering with the SYNTHETIC stand-in assets — RMSE against the reference's golden images is not meaningful; pass --assets")
        env = dict(os.environ, MI355PT_ASSETS=assets, MI355PT_DATA=os.path.join(ROOT, "toy-cpu-pathtracing_amd", "data"))
        for c, path in todo:
            out = os.path.join(tmp, c["output"])
            cmd = [a.cli, "--scene", str(c["scene"]), "--renderer", c["renderer"], "--sampler", c["sampler"], "--spp", str(c["spp"]),
                   "--width", str(c["width"]), "--height", str(c["height"]), "--output", out]
            r = subprocess.run(cmd, env=env, capture_output=True, text=True)
            entry = next(e for e in report if e["reference"] == c["reference"])
            if r.returncode != 0:
                failed += 1; entry["state"] = "render_failed"; entry["stderr"] = r.stderr[-400:]
                print(f"FAILED    {c['reference']}: renderer exit {r.returncode}: {r.stderr.strip()[-200:]}")
                continue
            got, ref = np.asarray(Image.open(out).convert("RGB")), np.asarray(Image.open(path).convert("RGB"))
            if got.shape != ref.shape:
                failed += 1; entry["state"] = "shape"; print(f"FAILED    {c['reference']}: {got.shape} vs {ref.shape}"); continue
            e = rmse_linear_u8(got, ref)
            entry["rmse"] = e; entry["state"] = "pass" if e <= c["max_rmse"] else "fail"
            failed += e > c["max_rmse"]
            print(f"{'PASS' if e <= c['max_rmse'] else 'FAIL'}      {c['reference']}: RMSE {e:.6f} (max {c['max_rmse']:.6f})")
    n = len(cases)
    print(f"{'unpinned' if absent or a.rehearsal else 'pinned'}: {absent}/{n} references absent, {len(todo)} compared, {failed} failed")
    if a.json:
        json.dump(report, open(a.json, "w"), indent=1)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
