#!/usr/bin/env python3
"""What a tail spill could save, estimated before any kernel was written: the oracle's per-pixel segment counts and hit
sequences of the headline frame give, for every workgroup of render_blocks (two 8x8 tiles) and every bounce, the rays
alive and the paths that hit; an instruction-cost model of a wave iteration (scan ~150 + 60 f + 45 trips, shading 350,
count barrier 25, pack 45) then prices the launch with the workgroups spilling at threshold T and a tail kernel shading
and scanning the handed-over paths in full waves (700 per 64 hits).  CPU only (oracle).

    python tools/tail_spill_model.py

Measured afterwards (profiles/r03_tail_spill.txt): render_blocks shrinks as modelled (0.71 of its time at T = 32, measured
0.76); the tail costs 2.5x the model's figure because it is latency-bound, not issue-bound."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import cases  # noqa: E402
import oracle  # noqa: E402
w, h = 1920, 1080
scene = cases.headline_scene(8)
sky, rnd = cases.inputs(256, 128)
fr = oracle.render(scene.live_object_bytes(), len(scene.objects), scene.config_bytes(w, h), (0.1, -0.2), 3, sky, rnd, threads=8)
seg = fr.segments.astype(np.int64); hits = fr.hits
print("S/P", seg.mean())
R = 8
# alive[k] : pixel runs scan k (1-based) ; hit[k]: scan k hit something
alive = np.stack([seg >= k for k in range(1, R + 1)], -1)
hit = (hits >= 0) & alive
# tiles: 8x8 ; WG = 2 tiles adjacent in x
th, tw = h // 8, w // 8
def per_tile(a):
    return a.reshape(th, 8, tw, 8, R).sum((1, 3))  # [th, tw, R]
A = per_tile(alive); H = per_tile(hit)
A2 = A.reshape(th, tw // 2, 2, R); H2 = H.reshape(th, tw // 2, 2, R)
def scan_cost(n, primary=False):
    if n == 0: return 0
    if primary: return 130
    return 150 + 60 * n / 64 + 45 * np.ceil(2.7 * n / 64)
SH, CNT, PACK, SPILL = 350, 25, 45, 40
def run(T, verbose=False):
    tot = 0.0; witer = 0; spilled = 0; tail_hits = 0
    for y in range(th):
        for x in range(tw // 2):
            a = A2[y, x]; hh = H2[y, x]
            live = 2
            # rays per wave at scan k
            rays = [a[0, 0], a[1, 0]]
            for k in range(R):
                n_alive = a[:, k].sum(); n_hit = hh[:, k].sum()
                if n_alive == 0: break
                if live == 2:
                    if k == 0: parts = [a[0, 0], a[1, 0]]
                    else: parts = [min(64, n_alive), max(0, n_alive - 64)] if packed else [pa, pb]
                else:
                    parts = [n_alive]
                for p in parts:
                    tot += scan_cost(p, k == 0) + CNT; witer += 1
                if n_hit == 0: break
                if k >= KMIN and n_hit <= T and k < R - 1:
                    tot += SPILL * live; spilled += n_hit
                    # tail work: all later hits of these paths
                    tail_hits += hh[:, k:].sum()
                    break
                need = (n_hit + 63) // 64
                packed = False
                if need < live:
                    tot += PACK * live; live = need; packed = True
                    tot += SH * live
                else:
                    tot += SH * live
                    # unpacked: waves keep own rays; approx split proportional
                    if live == 2:
                        pa, pb = a[0, k + 1] if k + 1 < R else 0, a[1, k + 1] if k + 1 < R else 0
                        if k > 0: # after an earlier pack identity lost; approximate split
                            nn = a[:, k + 1].sum() if k + 1 < R else 0
                            pa, pb = min(64, nn), max(0, nn - 64)
    tail = tail_hits / 64 * (SH + 70 + 280)
    return tot, tail, witer, spilled, tail_hits
KMIN = 1
base = run(0)
print("T=0 blocks cost %.1fM witer %d" % (base[0] / 1e6, base[2]))
for T in (8, 16, 24, 32, 48, 64):
    r = run(T)
    print("T=%d blocks %.1fM + tail %.1fM = %.1fM (%.3f of base) witer %d spilled %d (%.3f P) tail_hits %d" % (
        T, r[0] / 1e6, r[1] / 1e6, (r[0] + r[1]) / 1e6, (r[0] + r[1]) / base[0], r[2], r[3], r[3] / (w * h), r[4]))
KMIN = 0
print("spill allowed after the primary scan too")
for T in (16, 32, 48, 64, 96, 128):
    r = run(T)
    print("T=%d blocks %.1fM + tail %.1fM = %.1fM (%.3f of base) witer %d spilled %d (%.3f P)" % (
        T, r[0] / 1e6, r[1] / 1e6, (r[0] + r[1]) / 1e6, (r[0] + r[1]) / base[0], r[2], r[3], r[3] / (w * h)))
