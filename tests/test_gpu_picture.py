"""GPU: the HIP path against the reference's saved picture of its default scene, through the C ABI (tests/picture.py
says what the picture is and what the measures mean; tests/test_oracle_picture.py holds the same checks for the oracle
together with the controls that show how sharp they are)."""
import numpy as np
import pytest

import picture
from raytracer_amd import rtg

pytestmark = pytest.mark.gpu

W = picture.SIZE


def test_default_scene_matches_the_saved_picture(gpu_ctx):
    ref = picture.saved_picture()
    objs, lights, view = rtg.scene_builder(0)
    scene = gpu_ctx.scene([objs[0], objs[1]])
    # geometry: today's lights, no photon term
    _, c8, _ = scene.render_frame(view, lights, W, W)
    whole, ball, n, n_ball = picture.structure_agreement(ref, c8.reshape(W, W, 4)[:, :, :3].astype(np.float64))
    assert n > 120_000 and n_ball > 35_000
    assert whole > 0.955 and ball > 0.94, (whole, ball)
    # shading: the picture's lights, 2 x 2 000 000 photon directions (the picture had 2 x 10 000 000 of another seed)
    saved = picture.lights_of_the_saved_picture(lights)
    dirs = np.stack([rtg.fibonacci_zenith_y(2_000_000, seed=1 + i) for i in range(2)])
    photon_map = scene.photon_map(saved, dirs)
    assert photon_map.n > 20_000
    rgb, c8 = scene.render_frame_photons(view, saved, photon_map, W, W)
    rgb = rgb.reshape(W, W, 3)
    c8 = c8.reshape(W, W, 4)[:, :, :3].astype(np.float64)
    chroma, green, blue = picture.shading_agreement(ref, rgb, c8)
    assert chroma > 0.95 and green > 0.975 and blue > 0.955, (chroma, green, blue)
    whole, ball, _, _ = picture.structure_agreement(ref, c8)
    assert whole > 0.97 and ball > 0.965, (whole, ball)
    level_green, level_blue = picture.level_agreement(ref, rgb)
    assert level_green < 4.0 and level_blue < 5.0, (level_green, level_blue)
    direct, _, _ = scene.render_frame(view, saved, W, W)
    corr, off, n = picture.photon_term_agreement(ref, direct.reshape(W, W, 3), rgb)
    assert n > 15_000 and corr > 0.99 and off < 4.0, (corr, off, n)
    print(f"saved picture: structure {whole:.4f} (ball {ball:.4f}), green-blue balance {chroma:.4f}, "
          f"rank correlation green {green:.4f} blue {blue:.4f}, levels off by {level_green:.2f} / {level_blue:.2f} of 255 "
          f"(median), photon term alone: correlation {corr:.4f}, off by {off:.2f}, {photon_map.n} photons")
    photon_map.close()
    scene.close()
