"""Renders the default scene (SceneBuilder, optionally with another ball) with the photon term at 600x600 and writes
8-bit PNG files: for looking at next to the one picture the reference keeps of that scene (saved/sphere.jpg there).

  python tools/saved_picture.py OUT_DIR [radius ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import rtg  # noqa: E402


def main():
    from PIL import Image
    out = sys.argv[1]
    radii = [float(a) for a in sys.argv[2:]] or [0.2]
    os.makedirs(out, exist_ok=True)
    ctx = rtg.Context(0)
    w = h = 600
    for time in (0, 12):
        objs, lights, view = rtg.scene_builder(time)
        for r in radii:
            ball = rtg.copy_object(objs[0])
            ball.sphere_center[1] = -r
            ball.sphere_radius = r
            scene = ctx.scene([ball, objs[1]])
            dirs = np.stack([rtg.fibonacci_zenith_y(2_000_000, seed=1 + i) for i in range(len(lights))])
            pm = scene.photon_map(lights, dirs)
            rgb, c8 = scene.render_frame_photons(view, lights, pm, w, h)
            Image.fromarray(c8.reshape(h, w, 4)[:, :, :3]).save(os.path.join(out, f"t{time}_r{r}.png"))
            print(time, r, pm.n, float(rgb.max()), flush=True)
            pm.close()
            scene.close()


if __name__ == "__main__":
    main()
