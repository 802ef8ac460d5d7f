"""Renders the default scene as the reference's saved picture shows it (600x600, lights at (1, 1, 1) and (-1, 1, 1),
photon term) on the GPU, writes it as an 8-bit PNG through the sRGB curve and prints how it compares with the
picture (the measures of tests/picture.py).

  python tools/saved_picture.py OUT_DIR [photon directions per light]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import picture  # noqa: E402
from raytracer_amd import rtg  # noqa: E402


def main():
    from PIL import Image
    out = sys.argv[1]
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    os.makedirs(out, exist_ok=True)
    ref = picture.saved_picture()
    ctx = rtg.Context(0)
    w = picture.SIZE
    objs, lights, view = rtg.scene_builder(0)
    lights = picture.lights_of_the_saved_picture(lights)
    scene = ctx.scene([objs[0], objs[1]])
    dirs = np.stack([rtg.fibonacci_zenith_y(count, seed=1 + i) for i in range(len(lights))])
    photon_map = scene.photon_map(lights, dirs)
    rgb, c8 = scene.render_frame_photons(view, lights, photon_map, w, w)
    direct, _, _ = scene.render_frame(view, lights, w, w)
    rgb = rgb.reshape(w, w, 3)
    c8 = c8.reshape(w, w, 4)[:, :, :3].astype(np.float64)
    Image.fromarray(picture._srgb_levels(rgb).round().astype(np.uint8)).save(os.path.join(out, "default_scene_srgb.png"))
    print("photons", photon_map.n)
    print("structure (frame, ball, pixels, ball pixels)", picture.structure_agreement(ref, c8))
    print("green-blue balance, rank correlation green, blue", picture.shading_agreement(ref, rgb, c8))
    print("levels off by (green, blue; median, of 255)", picture.level_agreement(ref, rgb))
    print("photon term alone (correlation, levels off, pixels)", picture.photon_term_agreement(ref, direct.reshape(w, w, 3), rgb))
    photon_map.close()
    scene.close()


if __name__ == "__main__":
    main()
