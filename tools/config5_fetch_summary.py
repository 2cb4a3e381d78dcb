#!/usr/bin/env python3
"""profiles/r04_config5_fetch.txt from what `tools/collect_round4.sh d` leaves behind: the two traffic_r04_config5_sky*.json
(FETCH_SIZE / WRITE_SIZE passes) and the bench lines of the same passes (c5_<sky>_pmc0.json).

  python tools/config5_fetch_summary.py gpurun_out/prof_r04 > profiles/r04_config5_fetch.txt"""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r04"
rows = {}
for sky in (2048, 8192):
    t = json.load(open(os.path.join(d, "traffic_r04_config5_sky%d.json" % sky)))
    b = json.loads([l for l in open(os.path.join(d, "c5_%d_pmc0.json" % sky)) if l.startswith("{")][-1])
    r = b["roofline"]
    seg = r["kernel_mrays_per_s"] * r["avg_launch_ms"] * 1e3
    rows[sky] = dict(kernel=t["render_kernel"]["kernel"], fetch=t["render_kernel"]["fetch_bytes"], write=t["render_kernel"]["write_bytes"],
                     alg=r["alg_bytes_per_launch"], seg=seg, rate=b["value"] / 1e3, frames=b["config"]["frames_per_launch"],
                     w=b["config"]["width"] if "width" in b["config"] else 3840, h=b["config"].get("height", 2160))
a, c = rows[2048], rows[8192]
pixel_frames = 3840 * 2160 * a["frames"]
print("Config 5's frame (3840x2160, 64 spheres, 8 bounces, %d frames per launch) on one MI355X with the 32 MiB (2048x1024) and the 512 MiB "
      "(8192x4096)\nRGBA32F skybox: bytes the render step (%s) FETCHED and WROTE per launch (separate\nrocprofv3 --pmc FETCH_SIZE / "
      "WRITE_SIZE passes, one launch at a time; tools/collect_round4.sh d) against the ALGORITHMIC bytes of\nSURVEY.md 8d (8 B per pixel + "
      "16 B per segment + 8 B per rough hit).  Kernel sources: the round's final ones (sha256 in traffic_r04_config5_*.json).\n"
      % (a["frames"], a["kernel"]))
for sky, name in ((2048, "2048x1024 (32 MiB)"), (8192, "8192x4096 (512 MiB)")):
    r = rows[sky]
    print("skybox %s: fetched %.3f GB + written %.3f GB = %.3f GB per launch; algorithmic %.3f GB; segments per launch %.1f M"
          % (name, r["fetch"] / 1e9, r["write"] / 1e9, (r["fetch"] + r["write"]) / 1e9, r["alg"] / 1e9, r["seg"] / 1e6))
    print("    per segment: fetched %.1f B, written %.1f B, algorithmic %.1f B; fetched / algorithmic = %.2f; %.1f Grays/s under the counter pass"
          % (r["fetch"] / r["seg"], r["write"] / r["seg"], r["alg"] / r["seg"], r["fetch"] / r["alg"], r["rate"]))
extra = c["fetch"] - a["fetch"]
print("\nThe larger skybox costs %.2f GB of fetches more per launch = %.1f B per pixel-frame (%.1f M of them per launch): a path that ends on a "
      "miss looks\nthe skybox up once, in path_end — one 16-byte texel wanted, a 64-byte sector moved; neighbouring pixels' texels mostly share "
      "sectors, and with the\n32 MiB image most lookups hit in L2 / MALL where the 512 MiB image misses.  The written bytes are the float samples "
      "(12 B per pixel-frame) and the\npath queue of the tail spill." % (extra / 1e9, extra / pixel_frames, pixel_frames / 1e6))
