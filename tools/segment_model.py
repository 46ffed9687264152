"""Work model of the blend BACKWARD in two forms (CPU, numpy), on the bench scene - the evidence behind DESIGN.md 8.1 /
EXPERIMENTS.md round 5 ("tile-segment backward: modelled before building").

  (A) what ships: work item = 128 consecutive entries of ONE CELL LIST, four items of one length class per wave
      (rows of a wave are unrelated cells), 12 waves per CU; every (entry, cell) result leaves as a 40 B pair row that a
      second kernel sums per entry.
  (B) the tile-segment form: a workgroup of 4 waves owns a segment of consecutive tile-list
      entries of ONE tile (whole 64-record chunks, at most PMAX pairs: their 40 B results live in LDS), wave w takes
      four of the tile's 16 cells (cells sorted by the length of their sub-list, so the rows of a wave are of similar
      length), every row walks the part of its cell list that falls into the segment; after a workgroup barrier
      thread = entry adds its pairs in cell order and writes ONE 48 B row.  No pair rows in HBM, no second kernel.

Both forms are priced with the costs measured on the shipping kernel (profiles/r03_timeline.txt): a group
of four rows costs START + BATCH x (16-record batches of its longest row); START = 9.6 us, BATCH = 3.4 us when three
waves share a SIMD (a 1-batch group takes 13 us, an 8-batch group 37 us).  (B)'s workgroups per CU follow from its LDS
(4 x 8.7 KB of MFMA staging + 48 B per record of the segment + 40 B x PMAX of pair slots, of 160 KB), its batches are
priced by the issue share of the waves that fit (4 or 8 per CU against (A)'s 12) and its start at the same latency; a
workgroup ends with its slowest wave, its reduction is priced at 2 us.  Usage: python tools/segment_model.py [azim] [PMAX]"""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch

import oracle
from oracle import gs_oracle as go
from humangaussian_amd import synth

azim = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
PMAX = int(sys.argv[2]) if len(sys.argv) > 2 else 896
SEG = 128
START, BATCH = 9.6, 3.4
cloud = synth.init_cloud(100000, 0, "mid", seed=0)
cam = synth.orbit_camera(10.0, azim, 1.75, 55.0, 1024, 1024)
st = oracle.OracleSettings(1024, 1024, math.tan(cam.FoVx * .5), math.tan(cam.FoVy * .5), torch.zeros(3), 1.0,
                           cam.world_view_transform, cam.full_proj_transform, 0, cam.camera_center, False, False)
with torch.no_grad():
    pre = go.preprocess(cloud.means3D, None, cloud.shs, None, cloud.opacities, cloud.scales, cloud.rotations, None, st)
    g_sorted, t_sorted, ranges = go.bin_and_sort(pre)
gx = pre["grid"][0]
M2 = pre["mean2D"].numpy().astype(np.float64)
CON = pre["conic"].numpy().astype(np.float64)
OP = pre["opacity"].numpy().astype(np.float64)
gs, rg = g_sorted.numpy(), ranges.numpy()


def cell_masks(m, ca, cb, cc, op, x0t, y0t):
    tau = 2 * np.log(np.maximum(255 * op, 1.0))
    vis = 255 * op >= 1.0

    def qf(pxx, pyy):
        ddx = pxx - m[:, 0]
        ddy = pyy - m[:, 1]
        return ca * ddx * ddx + 2 * cb * ddx * ddy + cc * ddy * ddy
    out = []
    for cy in range(4):
        for cx in range(4):
            x0, y0 = x0t + cx * 4, y0t + cy * 4
            x1, y1 = x0 + 3, y0 + 3
            best = qf(np.clip(m[:, 0], x0, x1), np.clip(m[:, 1], y0, y1))
            for xe in (x0, x1):
                best = np.minimum(best, qf(xe, np.clip(m[:, 1] - (cb / cc) * (xe - m[:, 0]), y0, y1)))
            for ye in (y0, y1):
                best = np.minimum(best, qf(np.clip(m[:, 0] - (cb / ca) * (ye - m[:, 1]), x0, x1), ye))
            out.append(vis & (best <= tau))
    return np.stack(out, 1)


nb = lambda x: (x + 15) // 16                                         # noqa: E731
items_A = []                                                          # (A): entries of every work item
wgs_B = []                                                            # (B): per workgroup: batches of its 4 waves, rows used
pairs_total = entries_total = 0
for t in np.nonzero(rg[:, 1] > rg[:, 0])[0]:
    s, e = rg[t]
    gi = gs[s:e]
    n = e - s
    cm = cell_masks(M2[gi], CON[gi, 0], CON[gi, 1], CON[gi, 2], OP[gi], (t % gx) * 16.0, (t // gx) * 16.0)
    keep = cm.any(1)                                                  # (the rect cut drops entries that reach no cell)
    cm = cm[keep]
    n = cm.shape[0]
    if n == 0:
        continue
    entries_total += n
    pairs_total += int(cm.sum())
    for c in range(16):
        L = int(cm[:, c].sum())
        items_A += [SEG] * (L // SEG) + ([L % SEG] if L % SEG else [])
    # (B) segments: whole 64-record chunks, <= PMAX pairs (a chunk alone may exceed it: it is its own segment)
    chunk_pairs = [int(cm[k:k + 64].sum()) for k in range(0, n, 64)]
    k0 = 0
    while k0 < len(chunk_pairs):
        k1, acc = k0, 0
        while k1 < len(chunk_pairs) and (k1 == k0 or acc + chunk_pairs[k1] <= PMAX):
            acc += chunk_pairs[k1]
            k1 += 1
        sub = cm[k0 * 64:k1 * 64].sum(0)                              # records per cell inside the segment
        sub = np.sort(sub[sub > 0])[::-1] + (2 if k0 else 0)          # (+ ~2 context records behind a 4-aligned state)
        waves = [sub[w * 4:(w + 1) * 4] for w in range(4) if len(sub) > w * 4]
        wgs_B.append(([int(nb(w.max())) for w in waves], int(sum(nb(x) for x in sub)), int(len(sub))))
        k0 = k1

# ---- (A): groups of four items, longest first
items_A = np.sort(np.array(items_A))[::-1]
groups = [items_A[i:i + 4] for i in range(0, len(items_A), 4)]
gb = np.array([nb(g.max()) for g in groups])
rowb = sum(int(nb(x)) for x in items_A)
tA = (len(groups) * START + gb.sum() * BATCH)
slotsA = 256 * 12
print(f"azim {azim}: entries {entries_total}, pairs {pairs_total} ({pairs_total / entries_total:.2f} per entry)")
print(f"(A) items {len(items_A)}, groups {len(groups)} ({len(groups) / slotsA:.2f} per wave slot), wave-batches {gb.sum()}, "
      f"row-batch efficiency {rowb / (4 * gb.sum()):.3f}")
print(f"    wave time {tA / 1e3:.1f} ms -> perfectly balanced over {slotsA} wave slots: {tA / slotsA:.1f} us "
      f"(measured kernel 45 us; + pair_reduce 17.7 us = 62.6 us)")
# ---- (B)
wb = sum(sum(w) for w, _, _ in wgs_B)
nw = sum(len(w) for w, _, _ in wgs_B)
rb = sum(r for _, r, _ in wgs_B)
# workgroups per CU by LDS: 4 waves x 8.7 KB of MFMA staging + the segment's records (48 B x entries) + 40 B x PMAX pair slots
seg_entries = max(64, int(PMAX / (pairs_total / entries_total) / 64 + 1) * 64)
lds_wg = 4 * 8.7 * 1024 + 48 * seg_entries + 40 * PMAX
per_cu = max(1, int(160 * 1024 // lds_wg))
share = min(1.0, per_cu * 4 / 12.0)                                   # issue share of a batch relative to (A)'s 3 waves per SIMD
tB_wave = nw * START + wb * BATCH * share
wg_time = np.array([START + max(w) * BATCH * share + 2.0 for w, _, _ in wgs_B])      # workgroup = its slowest wave + reduction
slotsB = 256 * per_cu
print(f"    LDS per workgroup {lds_wg / 1024:.0f} KB -> {per_cu} workgroup(s) = {4 * per_cu} waves per CU")
print(f"(B) PMAX {PMAX}: workgroups {len(wgs_B)} ({len(wgs_B) / slotsB:.2f} per workgroup slot), waves {nw}, wave-batches {wb}, "
      f"row-batch efficiency {rb / (4 * wb):.3f}")
print(f"    wave time {tB_wave / 1e3:.1f} ms; workgroup time (slowest wave + reduction) {wg_time.sum() / 1e3:.1f} ms -> perfectly "
      f"balanced over {slotsB} workgroup slots: {wg_time.sum() / slotsB:.1f} us (no second kernel)")
