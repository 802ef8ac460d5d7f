"""GPU: configuration 5's frame -- 3840x2160, shaded to 8-bit colour, the 1 000 000-triangle terrain under the two
orbiting lights of SceneBuilder{time} (src/lib.rs:90-102) -- rendered whole and in tile mode (interleaved 8-row
strips, one rtg_gather_tiles through a real RCCL communicator on a side stream), for two orbit times; a strip of it
against the oracle."""
import os

import numpy as np
import pytest

import scenes
from raytracer_amd import rtg, shard

pytestmark = pytest.mark.gpu

W, H = 3840, 2160


@pytest.fixture(scope="module")
def one_rank_group():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group(backend="gloo", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


def test_4k_shaded_frame_whole_in_tiles_and_against_the_oracle(gpu_ctx, one_rank_group):
    import torch
    import bench
    ctx = gpu_ctx
    tris = rtg.heightfield(*bench.GRID, bench.SEED, bench.LOCAL_SCALE)
    bvh = rtg.HostBvh(tris)
    objs, _, view = scenes.mesh_scene_objects(rtg.transform(1.0 / bench.LOCAL_SCALE), with_default=False)
    pair = scenes.Pair(ctx, objs, {0: bvh})
    scene = pair.gpu
    ctx.set_option("comm_loopback", 2)                     # one rank, but every piece goes through ncclSend / ncclRecv
    comm = rtg.Comm(ctx, 1, 0, rtg.comm_unique_id())
    ctx.set_option("comm_loopback", 1)
    assert comm.info() == (1, 0, 1)
    side = torch.cuda.Stream()

    def gather(send, recv, pieces):
        comm.gather(0, send.data_ptr(), recv.data_ptr() if recv is not None else None, pieces,
                    stream_handle=side.cuda_stream)

    for time in (7, 31):
        _, lights, _ = rtg.scene_builder(time)
        lights = np.ascontiguousarray(lights, dtype=rtg.LIGHT_DTYPE)
        whole = torch.empty(W * H * 4, dtype=torch.uint8, device="cuda")
        scene.render_frame_raw(view, lights, len(lights), W, H, (0, 0, W, H), 4, None, whole.data_ptr(), None)
        ctx.synchronize()

        def render_rect(rect, out):
            scene.render_frame_raw(view, lights, len(lights), W, H, rect, 4, None, out.data_ptr(), None)
            ctx.synchronize()          # the gather runs on another stream than the renderer

        tiled = shard.render_frame_tiles(render_rect, W, H, 4, torch.uint8, "cuda", interleave=4, gather=gather)
        # read through torch straight away: the helper has ordered the gather's stream before it returns
        assert bool(torch.equal(tiled.reshape(-1), whole))
        lit = float((whole.view(H, W, 4)[:, :, :3].sum(dim=2) > 0).float().mean())
        assert lit > 0.5
        # rows 1000..1015 against the oracle: radiance bit for bit, colour within one level
        tile = (0, 1000, W, 16)
        g_rgb, g_c8, g_hits = scene.render_frame(view, lights, W, H, tile=tile, max_depth=4)
        c_rgb, c_c8, c_hits = pair.cpu.render_frame(view, lights, W, H, tile=tile, max_depth=4)
        assert scenes.hit_mismatches(g_hits, c_hits).size == 0
        assert not scenes.float_bits_differ(g_rgb, c_rgb).any()
        assert np.abs(g_c8.astype(int) - c_c8.astype(int)).max() <= 1
        strip = whole.view(H, W, 4)[1000:1016].cpu().numpy().reshape(-1, 4)
        assert np.array_equal(strip, g_c8)
    comm.close()
