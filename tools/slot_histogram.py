#!/usr/bin/env python3
"""Occupancy of the binned gather-dot (large offsets, k_gather_dot.hip) on a synthetic layer: how many sweeps per workgroup item the
window / slot assignment costs against the ideal (all 64 lanes of every sweep busy).

    python tools/slot_histogram.py [--offsets uniform|grid] [--S 256 --F 256 --G 9 --m 17 --radius 9 --windows 2]

Schemes (one sweep = 32 positions x one unit per lane; sweeps are counted per wave and item, 16 waves per workgroup):
  slot-pairs   round 2: wave = 4 input channels x slot pair gb; lane = (slot parity, f mod 32); a wave skips an input channel whose
               64 lane slots are all empty; a workgroup takes as long as its busiest wave
  half-wave    work list: one (input channel, slot index) per HALF wave (the MFMA's A operand is broadcast per half wave, CBSZ = 3),
               half-sweeps of all input channels dealt round robin over the 32 half waves of a workgroup
  quarter      the same with one (input channel, 16 output channels, slot index) per quarter wave (CBSZ = 2)
  packed       this round: a half-sweep = ANY 32 units of one (window, fb, input channel), dealt round robin so that an output
               channel's units go to different half-sweeps (its bank pair is read ceil(count / H) times per half-sweep: LDS
               bank conflicts instead of idle lanes); H = ceil(units / 32) half-sweeps per input channel
Writes one JSON line."""
import argparse, json
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--offsets", default="uniform")
ap.add_argument("--S", type=int, default=256); ap.add_argument("--F", type=int, default=256)
ap.add_argument("--G", type=int, default=9); ap.add_argument("--m", type=float, default=17.0)
ap.add_argument("--radius", type=int, default=9); ap.add_argument("--windows", type=int, default=2)
a = ap.parse_args()
rs = np.random.RandomState(0)
S, F, G = a.S, a.F, a.G
if a.offsets == "uniform":
    mu1 = rs.uniform(-a.m, a.m, (S, G, F)); mu2 = rs.uniform(-a.m, a.m, (S, G, F))
else:   # the layer's initial 3 x 3 grid +- 1 pixel per channel pair (bench.py --offsets grid)
    gx = int(np.ceil(np.sqrt(G))); g = np.arange(G)
    ax = np.arange(gx) * (2 * (a.m - 1) + 1) / gx + (-0.5 + (2 * (a.m - 1) + 1) / (2 * gx)) - (a.m - 1)
    mu1 = ax[g % gx].reshape(1, G, 1) + rs.uniform(-1, 1, (S, G, F)); mu2 = ax[g // gx].reshape(1, G, 1) + rs.uniform(-1, 1, (S, G, F))
R = a.radius * a.windows
w = lambda mu: np.minimum((np.floor(mu).astype(int) + R) // (2 * a.radius), a.windows - 1)
win = w(mu2) * a.windows + w(mu1)                                   # [S, G, F]
nwin = a.windows ** 2
cnt = np.stack([(win == i).sum(1) for i in range(nwin)])            # [window, S, F]: units of (s, f) in the window
nfb = F // 32
c = cnt.reshape(nwin, S, nfb, 32)
ideal = cnt.sum() / 64.0 / (nwin * nfb * (S // 64)) / 16.0          # sweeps per wave and item if every lane were busy
# round 2: slot pairs
kmax = c.max(3)                                                     # [window, S, fb]: slots the channel needs
pairs = (kmax + 1) // 2
sw = 0.0
ngb = (G + 1) // 2
for gb in range(ngb):
    act = (pairs > gb).reshape(nwin, S // 4, 4, nfb).sum(2)         # active channels of every wave
    wg = act.reshape(nwin, S // 64, 16, nfb).max(2)                 # busiest wave of the workgroup
    sw += wg.mean()
res = dict(offsets=a.offsets, S=S, F=F, G=G, windows=nwin, radius=a.radius, units_per_sf_window=float(cnt.mean()),
           ideal_sweeps_per_wave_item_64s=round(float(ideal), 3), slot_pairs_round2=round(float(sw), 3))
# work lists over all S of a (window, fb): sweeps per wave and item, normalised to 64 input channels like the others
half = kmax.sum(1).mean() / 32.0 / (S / 64.0)
q = c.reshape(nwin, S, nfb, 2, 16).max(4).sum((1, 3)).mean() / 64.0 / (S / 64.0)
e = c.reshape(nwin, S, nfb, 4, 8).max(4).sum((1, 3)).mean() / 128.0 / (S / 64.0)
res.update(half_wave_worklist=round(float(half), 3), quarter_wave_worklist=round(float(q), 3), eighth_wave_worklist=round(float(e), 3))
ns = c.sum(3)                                                       # [window, S, fb]: units of (window, s, fb)
H = (ns + 31) // 32
packed = H.sum(1).mean() / 32.0 / (S / 64.0)
mult = np.where(H > 0, -(-kmax // np.maximum(H, 1)), 0)             # worst bank-pair multiplicity of the channel's half-sweeps
res.update(packed_half_sweeps=round(float(packed), 3), packed_lane_fill=round(float(ns.sum() / (32.0 * H.sum())), 3),
           packed_worst_bank_multiplicity={int(k): int(v) for k, v in zip(*np.unique(mult, return_counts=True))})
res["slot_histogram"] = {int(k): int(v) for k, v in zip(*np.unique(cnt, return_counts=True))}
print(json.dumps(res))
