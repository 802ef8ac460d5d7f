import sys, time, numpy as np
sys.path.insert(0, '.')
from raytracer_amd import rtg
ctx = rtg.Context(0)
rgb, n, _ = rtg.render_scene_builder_photons(ctx, 0, 300, 300, 200000, seed=1)
def t(f, reps=50):
    f(); t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e3
print("to_hdr %.3f ms" % t(lambda: rtg.to_hdr(rgb, 300, 300)))
print("images_tar %.3f ms" % t(lambda: rtg.images_tar(rgb, 300, 300, seconds=0.01)))
