"""Where does the GPU idle inside a training step?  Reads a rocprofv3 --kernel-trace (+ --hip-runtime-trace) CSV pair of
`bench.py --no-cpu-baseline --no-extra` and prints, per kernel of the step, the mean gap between the previous kernel's end
and its start, next to the kernel durations, and the mean duration of the HIP calls the step makes.
  rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d DIR -o run -- python bench.py --no-cpu-baseline --no-extra --steps 100
  python tools/gap_trace.py DIR"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(kt)))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if "hgs_k_" in r["Kernel_Name"]))
# steps: from one hgs_k_preprocess_fwd to the next; the last 100 complete steps
starts = [i for i, k in enumerate(ks) if k[2].startswith("hgs_k_preprocess_fwd")]
starts = starts[-101:]
gap, dur, n = defaultdict(float), defaultdict(float), defaultdict(int)
span = 0.0
for a, b in zip(starts[:-1], starts[1:]):
    span += ks[b][0] - ks[a][0]
    for i in range(a, b):
        name = ks[i][2].split("(")[0]
        dur[name] += ks[i][1] - ks[i][0]
        gap[name] += ks[i][0] - ks[i - 1][1]
        n[name] += 1
steps = len(starts) - 1
print(f"steps {steps}: GPU step (preprocess_fwd start to the next) {span / steps / 1e3:.1f} us; kernels {sum(dur.values()) / steps / 1e3:.1f} us; gaps {sum(gap.values()) / steps / 1e3:.1f} us")
for name in dur:
    print(f"  {name:32s} dur {dur[name] / n[name] / 1e3:6.1f} us   idle before it {gap[name] / n[name] / 1e3:6.2f} us")
ht = sorted(glob.glob(d + "/**/*hip_api_trace.csv", recursive=True))
if ht:
    t = defaultdict(float); c = defaultdict(int)
    for r in csv.DictReader(open(ht[0])):
        t[r["Function"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); c[r["Function"]] += 1
    print("HIP calls (count, mean us):", ", ".join(f"{k} {c[k]} x {t[k] / c[k] / 1e3:.2f}" for k in sorted(t, key=lambda k: -t[k])[:8]))
