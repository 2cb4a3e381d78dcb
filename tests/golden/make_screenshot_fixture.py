#!/usr/bin/env python3
"""Turns the one output artefact the reference holds — screenshots/screenshot.png — into a test fixture.

Run in the build container (needs /root/reference; the GPU box never runs this):
    python tests/golden/make_screenshot_fixture.py

What the screenshot is (established here, checked again by tests/test_screenshot_pin.py):
  * The render panel of the 2552x1073 window shot is the rectangle x 5..1171, y 6..1056 (1167 x 1051 px).
    Inside it every 7th column and every 7th row repeats its neighbour exactly: egui shows the render
    target with a NEAREST sampler (render_target.rs:50-70) at 7/6 pixels per point.  Dropping the repeats
    leaves exactly 1000 x 900 texels — the reference's initial render size (app.rs:365) — i.e. the
    UNSCALED Rgba8UnormSrgb texels its fragment shader stored (PNG is lossless).
  * The panel beside it shows the camera: position 34.4 / 37.3 / 105.5, rotation 0.5 / -0.3 / 0, fov 70,
    bounces 3 — with ONE decimal (panels.rs:6-22 `fixed_decimals(1)`).  The position is used as shown; the two
    rotation angles are only known to +-0.05 rad (+-37 px), so they are fitted INSIDE that rounding interval by
    maximising the screenshot's edge strength along the silhouette edges of the oracle's primary-hit map.
Outputs (committed): tests/golden/reference_screenshot_panel.png (texel rows 0..559 of the 1000x900 target:
sky, horizon, every sphere; the rows below show only the ground) and reference_screenshot_panel.json."""
import json
import os
import sys

import numpy as np
from PIL import Image
from scipy import ndimage, optimize

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
import oracle  # noqa: E402
from rsraytracingv3_amd.scene import Camera, Scene  # noqa: E402

SHOT = "/root/reference/screenshots/screenshot.png"
PANEL = (5, 6, 1172, 1057)  # x0, y0, x1, y1 of the render panel in the window shot
ROWS_KEPT = 560
SHOWN = {"position": [34.4, 37.3, 105.5], "rotation": [0.5, -0.3, 0.0], "fov": 70.0, "bounces": 3}


def edge_strength(rgb_u8):
    sm = ndimage.gaussian_filter(rgb_u8.astype(np.float32), (1.2, 1.2, 0))
    gy, gx = np.zeros_like(sm), np.zeros_like(sm)
    gy[1:-1] = sm[2:] - sm[:-2]
    gx[:, 1:-1] = sm[:, 2:] - sm[:, :-2]
    return np.sqrt((gx ** 2 + gy ** 2).sum(-1))


def boundaries(ids):
    b = np.zeros(ids.shape, bool)
    d = ids[:, 1:] != ids[:, :-1]
    b[:, 1:] |= d
    b[:, :-1] |= d
    d = ids[1:] != ids[:-1]
    b[1:] |= d
    b[:-1] |= d
    return b


def main():
    shot = np.asarray(Image.open(SHOT).convert("RGB"))
    x0, y0, x1, y1 = PANEL
    reg = shot[y0:y1, x0:x1]
    d = reg.astype(int)
    dup_x = (np.abs(d[:, 1:] - d[:, :-1]).sum(-1) == 0).mean(0) > 0.9   # column i+1 repeats column i
    dup_y = (np.abs(d[1:] - d[:-1]).sum(-1) == 0).mean(1) > 0.9
    for dup in (dup_x, dup_y):  # one repeat per 7 displayed pixels (6 or 8 where the 7/6 ratio rounds the other way)
        assert set(np.diff(np.where(dup)[0]).tolist()) <= {6, 7, 8}
    keep_x = np.ones(reg.shape[1], bool)
    keep_x[np.where(dup_x)[0] + 1] = False
    keep_y = np.ones(reg.shape[0], bool)
    keep_y[np.where(dup_y)[0] + 1] = False
    texels = reg[keep_y][:, keep_x]
    assert texels.shape == (900, 1000, 3), texels.shape
    # every dropped row/column really was a copy — except under egui's resize grip in the bottom-right corner
    ix, iy = np.where(dup_x)[0], np.where(dup_y)[0]
    bad = np.argwhere(np.abs(d[:, ix + 1] - d[:, ix]).sum(-1) != 0)
    assert len(bad) < 20 and np.all(bad[:, 0] > reg.shape[0] - 14) and np.all(ix[bad[:, 1]] > reg.shape[1] - 14), bad
    bad = np.argwhere(np.abs(d[iy + 1] - d[iy]).sum(-1) != 0)
    assert len(bad) < 20 and np.all(iy[bad[:, 0]] > reg.shape[0] - 14) and np.all(bad[:, 1] > reg.shape[1] - 14), bad

    W, H = 1000, 900
    E = edge_strength(texels)
    scene = Scene.random_spheres_default_config()  # scene.rs:81-89, seed 42, on the reference's own random stream
    scene.reflection_limit = 1
    sky, rnd = cases.inputs(64, 32)
    objs = scene.as_bytes(W, H)[0]

    def score(q):
        scene.camera = Camera(tuple(SHOWN["position"]), (float(q[0]), float(q[1]), 0.0), SHOWN["fov"])
        ids = oracle.render(objs, 80, scene.config_bytes(W, H), (0.0, 0.0), 0, sky, rnd).hits[..., 0]
        return -float(E[boundaries(ids)].mean())

    lo = [SHOWN["rotation"][0] - 0.05, SHOWN["rotation"][1] - 0.05]
    hi = [SHOWN["rotation"][0] + 0.05, SHOWN["rotation"][1] + 0.05]
    grid = [(score((rx, ry)), rx, ry) for rx in np.linspace(lo[0], hi[0], 11) for ry in np.linspace(lo[1], hi[1], 11)]
    _, rx, ry = min(grid)
    start = np.array([rx, ry])
    res = optimize.minimize(score, start, method="Nelder-Mead", bounds=list(zip(lo, hi)),
                            options={"xatol": 1e-5, "fatol": 1e-3, "maxiter": 400,
                                     "initial_simplex": np.array([start, start + [0.004, 0], start + [0, 0.004]])})
    fitted = [round(float(res.x[0]), 5), round(float(res.x[1]), 5), 0.0]
    print("rotation fitted inside the rounding interval:", fitted, "edge score", -res.fun,
          "(score at the rounded values: %.1f)" % -score(SHOWN["rotation"][:2]))
    Image.fromarray(texels[:ROWS_KEPT]).save(os.path.join(HERE, "reference_screenshot_panel.png"), optimize=True)
    meta = {
        "source": "arlohb/RSRayTracingV3 screenshots/screenshot.png (2552x1073), panel x %d..%d y %d..%d, every 7th "
                  "row/column (nearest-sampler repeat at 7/6 px per point) dropped -> 1000x900 stored texels; rows 0..%d kept"
                  % (x0, x1 - 1, y0, y1 - 1, ROWS_KEPT - 1),
        "render_size": [W, H],
        "rows_kept": ROWS_KEPT,
        "shown": SHOWN,
        "camera": {"position": SHOWN["position"], "rotation": fitted, "fov": SHOWN["fov"]},
        "rotation_fit": "2 parameters, each inside shown +-0.05 (one displayed decimal); objective = mean edge strength of "
                        "the screenshot along the oracle's primary-hit silhouette edges (this script)",
        "edge_score": round(-float(res.fun), 2),
    }
    with open(os.path.join(HERE, "reference_screenshot_panel.json"), "w") as f:
        json.dump(meta, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
