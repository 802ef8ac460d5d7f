"""CPU: the output side of the frame API -- Radiance RGBE images (render/image.rs:33-71) and the tar archive the
reference's server frames them in (bin/server.rs:19-30)."""
import io
import tarfile

import numpy as np

from raytracer_amd import rtg


def _decode_hdr(data):
    head, _, rest = data.partition(b"\n\n")
    assert head.startswith(b"#?RADIANCE") and b"FORMAT=32-bit_rle_rgbe" in head
    dims, _, pixels = rest.partition(b"\n")
    _, h, _, w = dims.split()
    w, h = int(w), int(h)
    px = np.frombuffer(pixels, dtype=np.uint8)
    assert px.size == w * h * 4                      # flat (not run-length encoded) scanlines
    px = px.reshape(h, w, 4).astype(np.float64)
    scale = np.where(px[:, :, 3:] > 0, np.ldexp(1.0, px[:, :, 3:].astype(int) - 136), 0.0)
    return px[:, :, :3] * scale


def test_hdr_round_trip_within_rgbe_precision():
    rng = np.random.default_rng(5)
    rgb = rng.uniform(0.0, 4.0, size=(12, 20, 3)) * rng.choice([1e-3, 1.0, 50.0], size=(12, 20, 1))
    rgb[0, 0] = 0.0
    back = _decode_hdr(rtg.to_hdr(rgb, 20, 12))
    peak = rgb.max(axis=2, keepdims=True)
    assert np.all(np.abs(back - rgb) <= peak / 128.0 + 1e-300)


def test_images_tar_holds_the_three_images():
    rng = np.random.default_rng(6)
    rgb = rng.uniform(0.0, 2.0, size=(9, 16, 3))
    blob = rtg.images_tar(rgb, 16, 9, seconds=0.5)
    assert len(blob) % 512 == 0 and blob[-1024:] == bytes(1024)
    with tarfile.open(fileobj=io.BytesIO(blob)) as tar:
        members = {m.name: m for m in tar.getmembers()}
        assert sorted(members) == ["image.hdr", "image.pq.hdr", "perf.hdr"]
        for m in members.values():
            assert m.isfile() and m.mode == 0o700 and m.mtime == 0
        assert tar.extractfile(members["image.hdr"]).read() == rtg.to_hdr(rgb, 16, 9)
        perf = _decode_hdr(tar.extractfile(members["perf.hdr"]).read())
        want = 0.5 / (16 * 9) * 3000.0                # per-pixel time x 3000 (renderer.rs:152)
        assert np.all(np.abs(perf - want) <= want / 128.0)
        pq = _decode_hdr(tar.extractfile(members["image.pq.hdr"]).read())
        assert np.all((pq >= 0.0) & (pq <= 1.0))


def test_large_images_encode_like_their_rows():
    """Pictures above 65 536 pixels are encoded by several host threads (rows shared among them): the bytes equal
    those of the same rows encoded one strip at a time, NaN pixels (the reference's checkerboard, render/image.rs:52-58)
    included; the sizing calls agree with what is written."""
    rng = np.random.default_rng(8)
    w, h = 640, 411
    rgb = rng.uniform(0.0, 3.0, size=(h, w, 3)) * rng.choice([1e-4, 1.0, 700.0], size=(h, w, 1))
    rgb[rng.random((h, w)) < 0.01] = np.nan
    rgb[5, 7] = 0.0
    whole = rtg.to_hdr(rgb, w, h)
    head, _, rest = whole.partition(b"\n\n")
    pixels = rest.partition(b"\n")[2]
    strips = b""
    for y0 in range(0, h, 2):                      # two rows at a time keeps the checkerboard's parity
        piece = rtg.to_hdr(np.ascontiguousarray(rgb[y0:y0 + 2]), w, min(2, h - y0))
        strips += piece.partition(b"\n\n")[2].partition(b"\n")[2]
    assert pixels == strips
    blob = rtg.images_tar(rgb, w, h, seconds=1.25)
    with tarfile.open(fileobj=io.BytesIO(blob)) as tar:
        members = {m.name: m for m in tar.getmembers()}
        assert tar.extractfile(members["image.hdr"]).read() == whole
        pq = tar.extractfile(members["image.pq.hdr"]).read()
    small = rtg.images_tar(np.ascontiguousarray(rgb[:2]), w, 2)
    with tarfile.open(fileobj=io.BytesIO(small)) as tar:
        pq_small = tar.extractfile("image.pq.hdr").read()
    assert pq.partition(b"\n\n")[2].partition(b"\n")[2][: w * 2 * 4] == pq_small.partition(b"\n\n")[2].partition(b"\n")[2]


def test_fibonacci_sphere_directions():
    """Sphere::fibonacci_sphere + ZenithY::into_normal (geo/sphere.rs:27-32,48-80): unit vectors, heights climbing
    from -1 to +1 in steps of dydi = 2 / (count + d), zenith advancing by the golden angle."""
    n = 20000
    d = rtg.fibonacci_sphere(n, seed=3)
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-12)
    phi = np.pi * (3.0 - 5.0 ** 0.5)
    dd = 2.0 * np.pi / phi
    dydi = 2.0 / (n + dd)
    i = np.arange(n)
    y = d[:, 1]
    # y = -1 + dydi (i + u) + dydi d v with u, v in [0, 1)
    assert np.all(y >= -1.0 + dydi * i - 1e-15) and np.all(y < -1.0 + dydi * (i + 1 + dd))
    assert np.all(np.diff(y[::8]) > 0)
    # evenly spread: every octant holds about an eighth of the directions
    octant = (d[:, 0] > 0) * 4 + (d[:, 1] > 0) * 2 + (d[:, 2] > 0)
    assert np.abs(np.bincount(octant, minlength=8) / n - 0.125).max() < 0.01
    assert np.array_equal(d, rtg.fibonacci_sphere(n, seed=3)) and not np.array_equal(d, rtg.fibonacci_sphere(n, seed=4))
