#!/usr/bin/env python3
"""DESIGN.md 5.00's table from the files tools/final_evidence.sh wrote (gpurun_out/evidence or profiles/): prints markdown."""
import csv, json, os, sys
E = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/evidence"
c2 = json.loads(open(os.path.join(E, "bench_c2.json")).read().strip().splitlines()[-1])
others = [json.loads(l) for l in open(os.path.join(E, "bench_other_configs_1gpu.jsonl"))]
cols = [("C2", c2, "c2")] + [(n, o, k) for (n, k), o in zip((("C3", "c3"), ("C4", "c4"), ("C5", "c5")), others[:3])]
def kavg(k):
    rows = list(csv.DictReader(open(os.path.join(E, f"bench_{k}_kernel_stats.csv"))))
    r = [x for x in rows if "pt_kernel<false" in x["Name"]][0]
    return float(r["AverageNs"]) / 1e6, int(r["Calls"])
def row(name, f):
    print("| " + name + " | " + " | ".join(f(n, d, k) for n, d, k in cols) + " |")
row("Msamples/s", lambda n, d, k: f"**{d['value']:,.0f}**".replace(",", " "))
row("per step", lambda n, d, k: f"{d['ms_per_step']:,.1f} ms".replace(",", " "))
row("rocprofv3 kernel average", lambda n, d, k: "%.1f ms (%d calls)" % kavg(k))
row("wave-level VALU instructions per sample", lambda n, d, k: "%.1f" % d["roofline"]["valu"]["wave_instr_per_sample"])
row("VALU issue fraction = `roofline.frac`", lambda n, d, k: "%.3f" % d["roofline"]["frac"])
row("VALU lane use", lambda n, d, k: "%.3f" % d["roofline"]["valu"]["lane_use"])
row("useful lane-slots", lambda n, d, k: "%.3f" % d["roofline"]["valu"]["useful_lane_frac"])
row("L1 tag lookups per clock per CU", lambda n, d, k: "%.3f" % d["roofline"]["l1"]["tag_lookups_per_clk_per_cu"])
row("TA busy", lambda n, d, k: "%.2f" % d["roofline"]["l1"]["ta_busy"])
row("lookups per wave load / wave loads per sample", lambda n, d, k: "%.1f / %.2f" % (d["roofline"]["l1"]["lookups_per_wave_load"], d["roofline"]["l1"]["wave_loads_per_sample"]))
row("L1 hit / L2 hit", lambda n, d, k: "%.3f / %.3f" % (d["roofline"]["l1"]["l1_hit_rate"], d["roofline"]["l2_hit_rate"]))
row("L2 requests", lambda n, d, k: "%.2f TB/s" % (d["roofline"]["l2_request_GBps"] / 1e3))
row("HBM / fabric", lambda n, d, k: "%.0f GB per launch = %.0f GB/s = %.3f of peak" % (d["roofline"]["traffic"] / 1e9, d["roofline"]["hbm_GBps"], d["roofline"]["hbm_frac"]))
row("algorithmic B/sample, GB/s, / L2 requests", lambda n, d, k: "%.0f, %.0f, %.2f" % (d["roofline"]["algorithmic"]["bytes_per_sample"], d["roofline"]["algorithmic"]["GBps"], d["roofline"]["algorithmic"]["frac_of_l2_request_rate"]))
row("film check", lambda n, d, k: str((d.get("film_check") or {}).get("bit_exact")) + " / non-finite " + str((d.get("film_check") or {}).get("non_finite_values")))
print("wave shares C2:", c2["roofline"]["valu"]["wave_cycle_shares"])
print("cpu_baseline:", c2["cpu_baseline"]["value"], c2["cpu_baseline"]["cores"], "x", c2["value"] / c2["cpu_baseline"]["value"])
print("C1:", others[3]["ms_per_step"], others[3]["value"]); print("library:", c2["roofline"]["valu"]["profiled_library"])
