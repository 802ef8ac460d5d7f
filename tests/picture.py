"""Comparison of a rendered frame with the one picture the reference keeps of its default scene.

tests/golden/reference_saved_sphere.jpg is a copy of the reference's saved/sphere.jpg: a 600x600 JPEG of the glass
ball over the checker plane, photon term included, written by an earlier state of src/lib.rs whose two lights stood
at (1, 1, 1) (blue) and (-1, 1, 1) (green) -- the values the commented-out lights of lib.rs:98-100 still show; the
fit of the light position to the picture peaks there on all three axes.  The picture is lossy and its photon directions were seeded from entropy, so nothing
here is bit-exact: the first measures below are independent of any monotone tone curve and insensitive to photon noise, and sharp enough that a ball 2.5 % larger,
a refractive index of 1.45 instead of 1.5 or a frame shifted by one pixel fails them.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SIZE = 600
FLOOR_ROW = 340                       # rows below the horizon's aliased band
BALL = (300, 422, 118)                # the ball's disc in the picture: centre x, y and a radius just inside the rim

SAVED_LIGHTS = ((1.0, 1.0, 1.0), (-1.0, 1.0, 1.0))


def saved_picture():
    from PIL import Image
    return np.asarray(Image.open(os.path.join(HERE, "golden", "reference_saved_sphere.jpg")).convert("RGB")).astype(np.float64)


def _structure(img, win=11):
    from scipy import ndimage as ndi
    lum = img.sum(axis=2)
    mean = ndi.uniform_filter(lum, win)
    sd = np.sqrt(np.maximum(ndi.uniform_filter(lum * lum, win) - mean * mean, 0.0))
    return lum > mean, sd, mean


def structure_agreement(ref, ours):
    """Each pixel is classed as brighter or darker than its 11x11 neighbourhood -- the checker cells, the ball's rim,
    the refracted and reflected checker inside the ball -- wherever both pictures have local contrast; returns the
    fraction of such pixels classed alike (over the frame below the horizon, inside the ball) and the pixels counted."""
    a, sa, ma = _structure(ref)
    b, sb, mb = _structure(ours)
    mask = (sa > 0.15 * ma + 2) & (sb > 0.15 * mb + 2)
    mask[:330] = False
    yy, xx = np.mgrid[0:SIZE, 0:SIZE]
    ball = (xx - BALL[0]) ** 2 + (yy - BALL[1]) ** 2 < BALL[2] ** 2
    same = a == b
    return float(same[mask].mean()), float(same[mask & ball].mean()), int(mask.sum()), int((mask & ball).sum())


def _chroma(img, sigma=6):
    from scipy import ndimage as ndi
    g = ndi.gaussian_filter(img[:, :, 1], sigma)
    b = ndi.gaussian_filter(img[:, :, 2], sigma)
    return (g - b) / (g + b + 1e-3 * (g + b).max())


def shading_agreement(ref, rgb, rgb8):
    """Where the green and the blue light fall: correlation of the smoothed green-blue balance (of the 8-bit frames),
    and the rank correlation of the linear green and blue radiance with the picture's channels, over the floor rows."""
    from scipy import stats
    floor = np.zeros((SIZE, SIZE), bool)
    floor[FLOOR_ROW:] = True
    chroma = float(np.corrcoef(_chroma(ref)[floor], _chroma(rgb8)[floor])[0, 1])
    rank = [float(stats.spearmanr(ref[FLOOR_ROW:, :, c].ravel()[::7], rgb[FLOOR_ROW:, :, c].ravel()[::7]).statistic)
            for c in (1, 2)]
    return chroma, rank[0], rank[1]


def level_agreement(ref, rgb):
    """The picture's levels are the linear radiance through the sRGB transfer curve at unit scale (found by fitting: no
    other curve or scale comes close).  Returns the median absolute difference, in 8-bit levels, of the green and the
    blue channel between the picture and the frame encoded that way, over the floor rows, both smoothed over 3x3
    pixels.  An absolute check: the lights' intensity, the inverse-square and cosine factors, the two albedos of
    the checker and the flux a photon carries all scale it."""
    from scipy import ndimage as ndi
    encoded = _srgb_levels(rgb)
    out = []
    for c in (1, 2):
        d = ndi.uniform_filter(encoded[:, :, c], 3)[FLOOR_ROW:] - ndi.uniform_filter(ref[:, :, c], 3)[FLOOR_ROW:]
        out.append(float(np.median(np.abs(d))))
    return out[0], out[1]


def _srgb_levels(rgb):
    linear = np.clip(rgb, 0.0, 1.0)
    return 255.0 * np.where(linear <= 0.0031308, 12.92 * linear, 1.055 * linear ** (1.0 / 2.4) - 0.055)


def photon_term_agreement(ref, rgb_direct, rgb):
    """The photon term alone.  Inside the green light's shadow of the ball (direct green exactly zero in `rgb_direct`,
    the frame without the photon term; floor only, away from the ball and from the shadow's edge) all the green of the
    picture is light the ball focused there.  Returns the correlation and the median absolute difference in 8-bit
    levels between the picture's green and the frame's (sRGB-encoded, both smoothed over 5x5 pixels), and the pixels
    counted.  Green, because the picture's blue went through a colour conversion that its green did not (its red
    channel is not zero either, the scene's is)."""
    from scipy import ndimage as ndi
    yy, xx = np.mgrid[0:SIZE, 0:SIZE]
    region = (rgb_direct[:, :, 1] == 0) & (rgb_direct[:, :, 2] > 0) & (yy > FLOOR_ROW)
    region &= (xx - BALL[0]) ** 2 + (yy - BALL[1]) ** 2 > (BALL[2] + 12) ** 2
    region = ndi.binary_erosion(region, iterations=4)
    a = ndi.uniform_filter(ref[:, :, 1], 5)[region]
    b = ndi.uniform_filter(_srgb_levels(rgb)[:, :, 1], 5)[region]
    return float(np.corrcoef(a, b)[0, 1]), float(np.median(np.abs(a - b))), int(region.sum())


def lights_of_the_saved_picture(lights):
    out = lights.copy()
    for i, p in enumerate(SAVED_LIGHTS):
        out[i]["position"] = p
    return out
