"""Per-stage HBM-side traffic per step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 64 B per 128 B request for
wide coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated.  The counters
sit at the L2's fabric side, so Infinity-Cache hits are included: an upper bound on DRAM traffic."""
import collections, csv, json, sys

STAGE = [("hgs_k_preprocess_fwd", "preprocess_fwd"), ("hgs_k_tiles", "tiles"),
         ("hgs_k_fill", "fill"), ("hgs_k_sort", "sort"), ("hgs_k_render_fwd", "render_fwd"),
         ("hgs_k_render_bwd", "render_bwd"), ("hgs_k_pair_reduce", "pair_reduce"),
         ("hgs_k_preprocess_bwd", "preprocess_bwd")]


def load(path, counter):
    tot = collections.defaultdict(float)
    steps = 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        for pre, st in STAGE:
            if k.startswith(pre) or k.startswith("void " + pre):
                tot[st] += float(r["Counter_Value"])
                if pre == "hgs_k_preprocess_bwd":      # one per step (any instantiation)
                    steps += 1
                break
    return tot, max(steps, 1)


f, nf = load(sys.argv[1], "FETCH_SIZE")
w, nw = load(sys.argv[2], "WRITE_SIZE")
out = {st: (2.0 * f.get(st, 0.0) / nf + w.get(st, 0.0) / nw) * 1024.0 for st in sorted(set(f) | set(w))}
out["_steps"] = [nf, nw]
out["_commit"] = sys.argv[3] if len(sys.argv) > 3 else "unknown"      # the build the passes ran on
out["_note"] = ("per-step HBM-side bytes per stage = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate "
                "rocprofv3 --pmc passes (gfx950: FETCH_SIZE counts 64 B per 128 B request for wide "
                "coalesced reads; WRITE_SIZE uncalibrated); config 2; working set < 256 MiB Infinity "
                "Cache, so this is fabric traffic, an upper bound on DRAM traffic")
print(json.dumps(out, indent=1))
