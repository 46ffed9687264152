"""Per-kernel roofline table of DESIGN.md section 4 from the committed artifacts (profiles/r06_kernel_stats.csv,
profiles/r06_pmc_traffic.json, profiles/r06_bench.json): rewrites the block between the KTABLE markers and prints the
numbers the prose quotes.  python tools/design_numbers.py [--write]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
b = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
R, P, M, npix, T = b["config"]["num_rendered_R"], 100000, 1, 1 << 20, 4096
alg = {"preprocess_fwd": P * (44 + 12 * M + 76), "tiles": 8 * T, "fill": P * 16 + 12 * R, "sort": 24 * R,
       "render_fwd": 44 * R + 24 * npix, "render_bwd": 44 * R + 28 * npix, "pair_reduce": 40 * R,
       "preprocess_bwd": P * (60 + 12 * M + 44 + 12 * M) + P * 76 + 40 * R}
tr = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")))
dur = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r06_kernel_stats.csv")))}
names = {"preprocess_fwd": "hgs_k_preprocess_fwd", "tiles": "hgs_k_tiles", "fill": "hgs_k_fill", "sort": "hgs_k_sort_lds",
         "render_fwd": "hgs_k_render_fwd_store", "render_bwd": "hgs_k_render_bwd", "pair_reduce": "hgs_k_pair_reduce_em",
         "preprocess_bwd": "hgs_k_preprocess_bwd_s0"}
rows = ["| kernel | µs | algorithmic MB | % of 8 TB/s | counter MB | traffic ÷ algorithmic |", "|---|---|---|---|---|---|"]
tot_us = tot_tr = 0.0
for k in ("sort", "fill", "pair_reduce", "render_bwd", "preprocess_fwd", "render_fwd", "preprocess_bwd", "tiles"):
    n, us, a, t = names[k], dur[names[k]], alg[k] / 1e6, tr[k] / 1e6
    tot_us += us
    tot_tr += t
    rows.append(f"| `{n}` | {us:.1f} | {a:.1f} | {a / us / 8 * 100:.1f} | {t:.1f} | " + (f"{t / a:.1f}×" if a > 0.1 else "–") + " |")
path = (P * (292 + 36 * M) + 164 * R + 52 * npix + 8 * T) / 1e6
rows.append(f"| whole path | {tot_us:.1f} | {path:.1f} | {path / tot_us / 8 * 100:.1f} | {tot_tr:.1f} | {tot_tr / path:.1f}× |")
table = "\n".join(rows) + "\n"
print(table)
b1 = (84 * R + 28 * npix) / 1e6
b1us = dur[names["render_bwd"]] + dur[names["pair_reduce"]]
print(f"R {R}; B1 {b1:.1f} MB in {b1us:.1f} us = {b1 / b1us / 8 * 100:.1f} % (traffic {(tr['render_bwd'] + tr['pair_reduce']) / 1e6:.0f} MB); "
      f"step {b['ms_per_step']:.4f} ms, path frac {b['roofline']['path']['frac'] * 100:.1f} %; commit {tr.get('_commit')}")
for k, v in b["extra"].items():
    print(f"  {k}: {v['ms_per_step']:.4f} ms  {v['value'] / 1e6:.0f} M/s")
if "--write" in sys.argv:
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    s = re.sub(r"(<!-- KTABLE[^>]*-->\n).*?(<!-- /KTABLE -->)", lambda m: m.group(1) + table + m.group(2), s, flags=re.S)
    open(p, "w").write(s)
