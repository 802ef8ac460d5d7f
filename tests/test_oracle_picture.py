"""CPU: the oracle against the one output the reference holds for the ray-cast path, its saved picture of the default
scene (see tests/picture.py for what the picture is and what the measures mean).  Not a bit-exact pin -- the picture
is a lossy JPEG -- but every routine between render_pixel and the photon term has to be right for it to pass: the
controls at the end show what a small error does to the same measures."""
import numpy as np
import pytest

import orc
import picture
from raytracer_amd import rtg

W = picture.SIZE


def _render(lights, ball=None, photons=0, max_depth=4):
    objs, _, view = rtg.scene_builder(0)
    scene = orc.OracleScene([orc.oracle_object(ball if ball is not None else objs[0]), orc.oracle_object(objs[1])])
    if photons:
        dirs = np.stack([orc.fibonacci_zenith_y(photons, seed=1 + i) for i in range(len(lights))])
        found = scene.photons_emit(lights, dirs)
        kd = orc.OracleKd(found["position"])
        rgb, c8 = scene.render_frame_photons(view, lights, kd, found, W, W, max_depth=max_depth)
        kd.close()
    else:
        rgb, c8, _ = scene.render_frame(view, lights, W, W, max_depth=max_depth)
    scene.close()
    return rgb.reshape(W, W, 3), c8.reshape(W, W, 4)[:, :, :3].astype(np.float64)


@pytest.fixture(scope="module")
def ref():
    return picture.saved_picture()


def test_geometry_of_the_saved_picture(ref):
    """Primary rays, sphere, checker plane and the specular path tree (reflection and refraction to depth 4): the
    pattern of the picture, pixel for pixel, in the floor and inside the ball.  Independent of the lights."""
    _, lights, _ = rtg.scene_builder(0)
    _, c8 = _render(lights)
    whole, ball, n, n_ball = picture.structure_agreement(ref, c8)
    assert n > 120_000 and n_ball > 35_000
    assert whole > 0.955 and ball > 0.94, (whole, ball)
    # one pixel of shift is seen
    shifted, shifted_ball, _, _ = picture.structure_agreement(ref, np.roll(c8, 1, axis=1))
    assert shifted < 0.92 and shifted_ball < 0.90, (shifted, shifted_ball)


@pytest.mark.parametrize("what,change", [
    ("index 1.45", dict(n2=1.45)), ("index 1.55", dict(n2=1.55)),
    ("radius 0.195", dict(radius=0.195)), ("radius 0.205", dict(radius=0.205)),
    ("depth 2", dict(max_depth=2)),
])
def test_geometry_controls(ref, what, change):
    """The same measure rejects a slightly different ball and a shallower path tree."""
    objs, lights, _ = rtg.scene_builder(0)
    ball = rtg.copy_object(objs[0])
    if "n2" in change:
        ball.material.n2 = change["n2"]
    if "radius" in change:
        ball.sphere_radius = change["radius"]
        ball.sphere_center[1] = -change["radius"]
    _, c8 = _render(lights, ball=ball, max_depth=change.get("max_depth", 4))
    _, in_ball, _, _ = picture.structure_agreement(ref, c8)
    assert in_ball < 0.86, (what, in_ball)


def test_shading_of_the_saved_picture(ref):
    """Direct light with shadows plus the photon term, with the lights where the picture had them."""
    _, lights, _ = rtg.scene_builder(0)
    saved = picture.lights_of_the_saved_picture(lights)
    rgb, c8 = _render(saved, photons=1_000_000)
    chroma, green, blue = picture.shading_agreement(ref, rgb, c8)
    assert chroma > 0.95 and green > 0.975 and blue > 0.955, (chroma, green, blue)
    whole, ball, _, _ = picture.structure_agreement(ref, c8)
    assert whole > 0.955 and ball > 0.94, (whole, ball)
    # absolute levels: within a few 8-bit steps of the picture; lights a fifth weaker are not
    level_green, level_blue = picture.level_agreement(ref, rgb)
    assert level_green < 4.0 and level_blue < 5.0, (level_green, level_blue)
    dim_green, dim_blue = picture.level_agreement(ref, rgb * 0.8)
    assert dim_green > 6.0 and dim_blue > 7.0, (dim_green, dim_blue)
    # the photon term on its own, where no direct green light falls: right to a few levels; a quarter less flux is not
    rgb0, c80 = _render(saved)
    corr, off, n = picture.photon_term_agreement(ref, rgb0, rgb)
    assert n > 15_000 and corr > 0.99 and off < 4.0, (corr, off, n)
    _, off_less, _ = picture.photon_term_agreement(ref, rgb0, rgb0 + 0.75 * (rgb - rgb0))
    _, off_more, _ = picture.photon_term_agreement(ref, rgb0, rgb0 + 1.25 * (rgb - rgb0))
    assert off_less > 6.0 and off_more > 6.0, (off_less, off_more)
    # controls: without the photon term, and with a light a tenth of a unit away, the same measures drop
    chroma0, green0, blue0 = picture.shading_agreement(ref, rgb0, c80)
    assert green0 < green - 0.05 and blue0 < blue - 0.05, (green0, blue0)
    moved = saved.copy()
    moved[0]["position"] = (0.9, 1.0, 1.0)
    moved[1]["position"] = (-0.9, 1.0, 1.0)
    rgb1, c81 = _render(moved, photons=1_000_000)
    chroma1, green1, blue1 = picture.shading_agreement(ref, rgb1, c81)
    assert chroma1 < chroma - 0.04 and green1 < green - 0.01, (chroma1, green1, blue1)
