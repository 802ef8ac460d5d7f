"""CPU: the C-ABI library loads, exports every symbol include/rtg.h declares, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytracer_amd import rtg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_rtg_exports_every_declared_symbol():
    names = _declared("include/rtg.h", "rtg_")
    assert len(names) >= 17
    for n in names:
        assert hasattr(rtg.lib, n), f"librtg.so does not export {n}"


def test_the_documented_rust_binding_names_every_entry_point():
    """The Rust `extern "C"` block a maintainer of the reference would paste (kept with the integration notes at the
    repository root) declares exactly the functions of include/rtg.h: an entry point added to the header must reach
    the binding too."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    bound = set(re.findall(r"\bfn\s+(rtg_\w+)\s*\(", block))
    declared = set(_declared("include/rtg.h", "rtg_"))
    assert declared - bound == set(), f"not in the Rust binding: {sorted(declared - bound)}"
    assert bound - declared == set(), f"bound but not declared: {sorted(bound - declared)}"


def test_host_library_exports_every_declared_symbol():
    for n in _declared("raytracer_amd/host/rt_host_c.h", "rth_"):
        assert hasattr(rtg.host, n), f"librt_host.so does not export {n}"


def test_struct_sizes_match_the_header():
    # offsets are checked by the parity tests; sizes catch padding mistakes early
    assert C.sizeof(rtg.BvhNode) == 64 and C.sizeof(rtg.Bvh) == 48
    assert C.sizeof(rtg.Object) == 8 + 328 + 48 + 48 + 24 + 8 + 72 + 8
    assert C.sizeof(rtg.View) == 40 and C.sizeof(rtg.Light) == 48 and C.sizeof(rtg.Rect) == 16
    assert rtg.POINT_DTYPE.itemsize == 112 and rtg.COUNTER_DTYPE.itemsize == 8


def test_version_string():
    assert rtg.version().startswith("rtg ") and "gfx950" in rtg.version()


def test_null_arguments_are_errors_not_crashes():
    assert rtg.lib.rtg_ctx_create(0, None) == -1
    assert b"null" in rtg.lib.rtg_last_error(None)
    assert rtg.lib.rtg_ctx_synchronize(None) == -1
    assert rtg.lib.rtg_mesh_create(None, None, None) == -1
    assert rtg.lib.rtg_scene_create(None, None, None) == -1
    assert rtg.lib.rtg_render_primary(None, None, 1, 1, rtg.Rect(0, 0, 1, 1), None, None, None) == -1
    assert rtg.lib.rtg_trace(None, None, 0, None, None, None) == -1
    n = C.c_uint64(7)
    assert rtg.lib.rtg_photons_emit(None, None, 0, None, 0, 4, None, 0, C.byref(n)) == -1
    assert rtg.lib.rtg_photon_map_create(None, None, 0, None) == -1
    assert rtg.lib.rtg_photon_map_emit(None, None, 0, None, 0, 4, None, None) == -1
    assert rtg.lib.rtg_photon_map_nearest(None, None, 0, 1, None, None) == -1
    assert rtg.lib.rtg_photon_map_order(None, None) == -1
    assert rtg.lib.rtg_indirect_irrad(None, None, None, 0, 3, 5, 1e-5, None) == -1
    assert rtg.lib.rtg_render_frame_photons(None, None, None, 0, None, 3, 5, 1e-5, 1, 1, rtg.Rect(0, 0, 1, 1), 4, None, None) == -1
    assert rtg.lib.rtg_render_primary_static(None, None, 1, 1, rtg.Rect(0, 0, 1, 1), None) == -1
    assert rtg.lib.rtg_gather_tiles(None, 0, None, None, None, 0, None) == -1
    assert rtg.lib.rtg_comm_create(None, None, 1, 0, None) == -1
    assert rtg.lib.rtg_comm_info(None, None, None, None) == -1
    rtg.lib.rtg_ctx_destroy(None)
    rtg.lib.rtg_mesh_destroy(None)
    rtg.lib.rtg_scene_destroy(None)
    rtg.lib.rtg_photon_map_destroy(None)
    rtg.lib.rtg_comm_destroy(None)


def _has_gpu():
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device behaviour")
def test_no_device_fails_loudly():
    with pytest.raises(rtg.RtgError) as e:
        rtg.Context(0)
    assert e.value.code == -4 and "no HIP device" in str(e.value)


def test_host_generators_are_deterministic():
    a = rtg.heightfield(8, 4, seed=3, scale=2.0)
    b = rtg.heightfield(8, 4, seed=3, scale=2.0)
    c = rtg.heightfield(8, 4, seed=4, scale=2.0)
    assert a.tobytes() == b.tobytes() and a.tobytes() != c.tobytes()
    # no triangle has a zero-thickness box (they would be invisible to the slab test)
    ext = a.max(axis=1) - a.min(axis=1)
    assert (ext > 0).all()
    box = [0, 0, 0, 1, 2, 3]
    r = rtg.bench_mesh_rays(box, 1000, seed=5)
    assert r.shape == (1000, 6) and (r[:, :3] >= 0).all() and (r[:, :3] <= [1, 2, 3]).all()
    assert np.allclose(np.linalg.norm(r[:, 3:], axis=1), 1.0, atol=1e-15)
    perm, s = rtg.sort_rays(box, r)
    assert sorted(perm.tolist()) == list(range(1000)) and np.array_equal(s, r[perm])


def test_scene_builder_defaults():
    objs, lights, view = rtg.scene_builder(0)
    assert objs[0].kind == rtg.SPHERE and objs[1].kind == rtg.PLANE
    assert list(objs[0].sphere_center) == [0.0, -0.2, 0.0] and objs[0].sphere_radius == 0.2
    assert objs[0].material.has_dielectric == 1 and (objs[0].material.n1, objs[0].material.n2) == (1.0, 1.5)
    assert list(objs[1].plane_position) == [0.0, -0.5, 0.0] and list(objs[1].material2.diffuse) == [0.1, 0.1, 0.1]
    assert list(lights[0]["position"]) == [1.0, 0.0, 1.0] and list(lights[0]["color"]) == [0.0, 0.0, 100.0]
    assert list(lights[1]["position"]) == [-1.0, 0.0, 1.0] and list(lights[1]["color"]) == [0.0, 100.0, 0.0]
    assert list(view.orig) == [0.0, 0.0, 1.0]
    _, l25, _ = rtg.scene_builder(25)
    assert abs(l25[0]["position"][1] - 1.0) < 1e-15


def test_header_is_plain_c_and_layouts_match_the_python_mirror(tmp_path):
    """include/rtg.h compiles as C11 on its own (no C++ or framework types in the boundary), and the record sizes a
    C caller sees are the ones the ctypes / numpy mirrors use."""
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text(
        '#include "rtg.h"\n#include <stdio.h>\n#include <stddef.h>\n'
        'int main(void) {\n'
        '  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(rtg_bvh_node), sizeof(rtg_tri), sizeof(rtg_hit),\n'
        '         sizeof(rtg_point), sizeof(rtg_path), offsetof(rtg_path, point), sizeof(rtg_light), sizeof(rtg_view),\n'
        '         sizeof(rtg_patch_leaf), sizeof(rtg_trispline));\n'
        '  return 0;\n}\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [rtg.NODE_DTYPE.itemsize, 72, rtg.HIT_DTYPE.itemsize, rtg.POINT_DTYPE.itemsize, rtg.PATH_DTYPE.itemsize,
                   rtg.PATH_DTYPE.fields["point"][1], rtg.LIGHT_DTYPE.itemsize, C.sizeof(rtg.View),
                   rtg.PATCH_LEAF_DTYPE.itemsize, 240]
    assert rtg.PATH_DTYPE.itemsize == 144
