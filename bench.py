#!/usr/bin/env python3
"""bench.py -- Mrays/s of the GPU ray-cast path on the 1M-triangle BVH scene.

A step is one pass of the hot path over one batch: a full 1920x1080 primary-ray frame (ray generation +
BVH traversal + ray/triangle intersection + closest hit -> 16-byte hit records) against the 1 000 000-triangle
terrain, inputs (scene, BVH) resident in HBM, outputs written to HBM.  With N > 1 every rank renders its own
frame of the sequence (frame f -> rank f mod N, camera moved per frame; BVH replicated) and the hit buffers are
gathered to rank 0 inside the timed region by the library's own gather (rtg_gather_tiles: RCCL point-to-point
transfers behind the C ABI), on a second stream so the gather of frame i overlaps the traversal of frame i+1 (two hit
buffers per rank): weak scaling.  Every GPU keeps `--inflight` frames in flight (default 2: two contexts with a
stream each take the frames in turn, sharing one uploaded tree), because the tail of a persistent launch -- its last
rays finishing on a half-empty GPU -- is best hidden under the next frame's start.

The timed region is `--windows` windows of exactly `--steps` steps, each bracketed by a barrier and a device
synchronize on both sides; `value` / `ms_per_step` are the median window's, `windows_mrays` lists them all.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (k_trace_phased, the persistent traversal
kernel behind rtg_render_primary):
  * the bound it is held against is the one that binds a tree that sits in the caches (config 2: 144 MB of nodes and
    triangles inside the 256 MB Infinity Cache): the rate at which a CU's vector memory path serves gathered 16-byte
    lane loads, measured by tools/micro/tcp_gather.hip per cache level and weighted by the kernel's own hit fractions
    (profiles/<round>/ceiling.json, written by tools/roofline_ceiling.py from the committed microbenchmark output
    and counter passes).  `achieved` = node visits per launch (counted by the kernel's counting build on the same
    frame) / the launch's duration, `peak` = that ceiling, `frac` = achieved / peak;
  * `traffic` is the HBM traffic per launch from the rocprofv3 counter passes committed under profiles/
    (2 x FETCH_SIZE + WRITE_SIZE) and `hbm` the HBM roofline from it (traffic / duration against 8 TB/s);
  * `algorithmic` is nodes visited x 64 B + triangles tested x 72 B + 16 B hit record per ray over the duration -- bytes
    the traversal asks for, almost all served by L1/L2/Infinity Cache, so a figure above HBM peak is not a fraction
    of anything and is reported for information only.
The launch duration comes from HIP events that the launch itself fills with its execution window on the launching
stream; with two launches in flight a launch spans the time it shares the GPU with its neighbour
(`concurrent_launches`).  `configs.c4` is the same for configuration 4 (64 M incoherent rays against 10 M triangles,
the HBM-resident case), `configs.c3` for configuration 3 (3840x2160 camera rays against 15 500 trispline patches) and
`configs.c5` is configuration 5 on this GPU (the 3840x2160 frame sequence shaded to 8-bit colour under the orbiting
lights).  `cpu_baseline` is the oracle (C++ restatement of the reference renderer, kind "port") timed on this host's
cores on whole frames of the same sequence.

With N > 1 the line also carries `gather`: the rank count RCCL's communicator reports, bytes gathered per step, the
gather's own duration and rate per sending rank, and the time the traversal streams spent waiting for a hit buffer
whose gather had not finished.  A gather that cannot be set up natively ends the run with a non-zero exit (unless
`--gather torch` asked for the torch path), and so does a first gather that has not completed after
`--gather-timeout` seconds (what is pending is printed first).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
WIDTH, HEIGHT = 1920, 1080
GRID = (1000, 500)             # quads -> 1 000 000 triangles
SEED = 7
LOCAL_SCALE = 100.0
C4_GRID = (3163, 1581)         # quads -> 10 001 406 triangles
C4_RAYS = 1 << 26
C4_RAY_SEED = 10212233         # the reference benchmark's seed constant (src/tree/benchmark.rs:15)


def frame_view(rtg, frame):
    """Camera of frame `frame` of the sequence: the default view dollying slowly sideways."""
    return rtg.make_view(orig=(0.002 * frame, 0.0, 1.0))


KERNEL_OF_TRAVERSAL = {-1: "k_trace_phased", 0: "k_render_primary", 1: "k_render_primary", 2: "k_trace_phased"}


def committed(name, workload_key, kernel):
    """The newest committed profiles/<dir>/<name stem>*.json for this workload and kernel, or None."""
    best = None
    prof = os.path.join(ROOT, "profiles")
    if not os.path.isdir(prof):
        return None
    stem = name[:-len(".json")]
    for d in sorted(os.listdir(prof)):
        if not os.path.isdir(os.path.join(prof, d)):
            continue
        for fn in sorted(os.listdir(os.path.join(prof, d))):
            if not (fn.startswith(stem) and fn.endswith(".json")):
                continue
            with open(os.path.join(prof, d, fn)) as f:
                t = json.load(f)
            if t.get("workload") == workload_key and t.get("kernel") == kernel:
                t["file"] = os.path.join("profiles", d, fn)
                best = t
    return best


def roofline_record(kernel, workload_key, kernel_ms, n_rays, nodes_per_ray, tris_per_ray, ray_in_bytes, use_committed,
                    tree=None, leaf_bytes=72.0):
    """The `roofline` object.  Two ceilings are priced -- `gather` (the rate at which the CUs' vector memory path
    serves the node fetches, per cache level) and `hbm` (counter traffic against HBM peak) -- and the top-level
    bound / achieved / peak / frac are those of the one that binds (the larger fraction).  `algorithmic` is the byte
    rate the traversal asks for, for information; `compulsory` (given `tree` = (nodes, leaves)) is the least a launch
    could move: the scene as the kernel stores it, once (32-byte hot nodes, 80-byte triangle records), the rays in and
    the hit records out.  `little` is the latency ceiling: with one node fetch in flight per resident ray, no more
    than (rays resident on the chip) / (measured latency of a vector-memory instruction) visits can complete per
    second, whatever the caches deliver."""
    seconds = kernel_ms * 1e-3
    bytes_per_ray = nodes_per_ray * 64.0 + tris_per_ray * leaf_bytes + ray_in_bytes + 16.0
    visits_per_s = nodes_per_ray * n_rays / seconds
    ceiling = committed("ceiling.json", workload_key, kernel) if use_committed else None
    traffic = committed("traffic.json", workload_key, kernel) if use_committed else None
    gather = {"bound": "vector-memory gather rate (node fetches served by L1 / L2 / beyond)",
              "achieved": round(visits_per_s / 1e9, 2), "peak": None, "unit": "G node visits/s", "frac": None}
    if ceiling is not None:
        gather["peak"] = round(ceiling["ceiling_visits_per_s"] / 1e9, 2)
        gather["frac"] = round(visits_per_s / ceiling["ceiling_visits_per_s"], 4)
        gather["source"] = ceiling["file"]
    hbm = None
    if traffic is not None:
        b = float(traffic["hbm_bytes_per_launch"])
        hbm = {"bound": "hbm", "achieved": round(b / seconds / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(b / seconds / 1e9 / HBM_PEAK_GBS, 4), "source": traffic["file"],
               "what": "L2 -> fabric traffic per launch (2 x FETCH_SIZE + WRITE_SIZE; reads served by the Infinity "
                       "Cache are counted, so this is an upper bound on HBM utilisation: no counter on this GPU "
                       "splits them -- TCC_EA0_RDREQ_DRAM equals TCC_EA0_RDREQ) against the 8 TB/s HBM peak"}
    binding = hbm if (hbm is not None and (gather["frac"] is None or hbm["frac"] > gather["frac"])) else gather
    rec = {k: binding[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
    rec.update({"traffic": None if traffic is None else round(float(traffic["hbm_bytes_per_launch"]) / 1e9, 4),
                "traffic_unit": "GB per launch", "kernel": kernel, "kernel_ms": round(kernel_ms, 4),
                "nodes_per_ray": round(float(nodes_per_ray), 2), "tris_per_ray": round(float(tris_per_ray), 2),
                "gather": gather, "hbm": hbm, "little": little_record(kernel, workload_key, visits_per_s, use_committed),
                "algorithmic": {"bytes_per_ray": round(bytes_per_ray, 1),
                                "gbs": round(bytes_per_ray * n_rays / seconds / 1e9, 1),
                                "note": "bytes the traversal asks for; mostly cache hits, not HBM traffic"}})
    if tree is not None:
        least = tree[0] * 32.0 + tree[1] * 80.0 + n_rays * (ray_in_bytes + 16.0)
        rec["compulsory"] = {"bytes_per_launch": round(least / 1e9, 4), "unit": "GB",
                             "traffic_over_compulsory": None if traffic is None else
                             round(float(traffic["hbm_bytes_per_launch"]) / least, 2),
                             "note": "scene once (32-byte hot nodes, 80-byte triangles) + rays in + hit records out"}
    return rec


def little_record(kernel, workload_key, visits_per_s, use_committed):
    """Little's-law ceiling of a kernel whose rays each keep one dependent node fetch in flight: resident rays /
    measured vector-memory latency (profiles/<round>/latency*.json, from the SQ counters of tools/prof_latency.sh)."""
    lat = committed("latency.json", workload_key, kernel) if use_committed else None
    if lat is None:
        return None
    rays_resident = 1024 * lat["waves_per_simd"] * 64                    # 256 CUs x 4 SIMDs, 64 lanes per wave
    ceiling = rays_resident / (lat["vmem_latency_cycles"] / (lat["clock_ghz"] * 1e9))
    return {"bound": "latency (Little's law): resident rays / vector-memory latency", "unit": "G node visits/s",
            "achieved": round(visits_per_s / 1e9, 2), "peak": round(ceiling / 1e9, 1),
            "frac": round(visits_per_s / ceiling, 4), "rays_resident": rays_resident,
            "vmem_latency_cycles": lat["vmem_latency_cycles"], "clock_ghz": lat["clock_ghz"], "source": lat["file"],
            "note": "a wave waits for the slowest of its lanes and issues ~50 instructions per step as well, so a "
                    "step takes two to three such latencies; the ceiling prices the latency alone"}


def run_config3(rtg, torch, ctx, dev, launches=5):
    """Configuration 3 on one GPU: 3840x2160 camera rays against 15 500 trispline patches (992 000 facets in the
    tree, Newton refinement on the patch), hit records only.  Returns the `configs.c3` record."""
    W, H = 3840, 2160
    t0 = time.time()
    patches = rtg.patch_terrain(125, 62, seed=SEED, scale=LOCAL_SCALE)
    pm = rtg.HostPatchMesh(patches, k=8, newton_steps=5)
    mesh = ctx.mesh(pm)
    scene = ctx.scene([rtg.patch_object(mesh, rtg.material((1, 1, 1)), rtg.transform(1.0 / LOCAL_SCALE))])
    view = rtg.make_view()
    n = W * H
    tile = (0, 0, W, H)
    hits = torch.empty(n * 16, dtype=torch.uint8, device=dev)
    ctr = torch.empty(n * 2, dtype=torch.int32, device=dev)
    scene.render_primary_raw(view, W, H, tile, hits.data_ptr(), None, ctr.data_ptr())
    ctx.synchronize()
    setup_s = time.time() - t0
    c = ctr.view(n, 2).to(torch.int64).sum(dim=0).cpu().numpy()
    del ctr
    scene.render_primary_raw(view, W, H, tile, hits.data_ptr())
    ctx.synchronize()
    ctx.set_option("time_kernels", 2)
    ctx.kernel_time()
    t0 = time.perf_counter()
    for _ in range(launches):
        scene.render_primary_raw(view, W, H, tile, hits.data_ptr())
    ctx.synchronize()
    wall = time.perf_counter() - t0
    ms, k = ctx.kernel_time()
    ctx.set_option("time_kernels", 1)
    assert k == launches
    n_hit = int((hits.view(torch.int32).view(n, 4)[:, 1] != -1).sum().item())
    rec = {"workload": f"config 3: {W}x{H} primary rays vs {len(patches)} trispline patches ({pm.bvh.n_leaves} facets, "
                       f"{pm.bvh.n_nodes} nodes, depth {pm.bvh.depth}), closest-hit ids",
           "value": round(n * launches / wall / 1e6, 1), "unit": "Mrays/s", "launches": launches,
           "ms_per_launch": round(wall / launches * 1e3, 3), "rays_hit": n_hit, "setup_s": round(setup_s, 2),
           "roofline": roofline_record("k_trace_phased", "config3-3840x2160-patches", ms / launches, n, c[0] / n, c[1] / n,
                                       0.0, True, (pm.bvh.n_nodes, pm.bvh.n_leaves), leaf_bytes=72.0 + 240.0)}
    scene.close()
    mesh.close()
    return rec


def run_config5(rtg, torch, ctx, dev, scene, frames=24):
    """Configuration 5 on one GPU: a sequence of 3840x2160 frames of the 1 000 000-triangle terrain, shaded to 8-bit
    colour under the two lights of SceneBuilder{time} (src/lib.rs:90-102), frame f at time f (the loop of
    src/bin/local.rs:42-50).  With several GPUs the frames go to the ranks in turn and every frame's payload is
    gathered (tools/config5.py); here the one GPU renders them all.  Returns the `configs.c5` record."""
    W, H = 3840, 2160
    out = [torch.empty(W * H * 4, dtype=torch.uint8, device=dev) for _ in range(2)]
    lights = []
    for f in range(frames):
        _, l, view = rtg.scene_builder(f)
        lights.append(np.ascontiguousarray(l, dtype=rtg.LIGHT_DTYPE))

    def frame(f):
        scene.render_frame_raw(view, lights[f], len(lights[f]), W, H, (0, 0, W, H), 4, None, out[f & 1].data_ptr(), None)

    frame(0)
    ctx.synchronize()
    t0 = time.perf_counter()
    for f in range(frames):
        frame(f)
    ctx.synchronize()
    wall = time.perf_counter() - t0
    lit = float((out[(frames - 1) & 1].view(H, W, 4)[:, :, :3].sum(dim=2) > 0).float().mean())
    return {"workload": f"config 5 on one GPU: {frames} frames {W}x{H}, primary + shadow rays of two orbiting lights + "
                        f"direct shading + PQ 8-bit colour, 1 000 000-triangle terrain",
            "value": round(frames / wall, 2), "unit": "frames/s", "ms_per_frame": round(wall / frames * 1e3, 3),
            "mpixels_per_s": round(frames * W * H / wall / 1e6, 1), "payload_bytes_per_frame": W * H * 4,
            "lit_pixels": round(lit, 3)}


def run_config4(rtg, torch, ctx, dev, launches=3):
    """Configuration 4 on one GPU: 67 108 864 `bench_mesh` rays against the 10 001 406-triangle tree (built on the
    GPU), hit records only.  Returns the `configs.c4` record."""
    t0 = time.time()
    tris = rtg.heightfield(C4_GRID[0], C4_GRID[1], SEED, 1.0)
    mesh = ctx.build_mesh(tris)
    n_nodes, n_leaves, _, depth = mesh.sizes()
    scene = ctx.scene([rtg.mesh_object(mesh, rtg.material((1, 1, 1)), rtg.transform())])
    rays = rtg.bench_mesh_rays(mesh.bounds(), C4_RAYS, seed=C4_RAY_SEED)
    d_rays = torch.from_numpy(rays).to(dev)
    del rays
    hits = torch.empty(C4_RAYS * 16, dtype=torch.uint8, device=dev)
    ctr = torch.empty(C4_RAYS * 2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    setup_s = time.time() - t0
    scene.trace_raw(d_rays.data_ptr(), C4_RAYS, hits.data_ptr(), None, ctr.data_ptr())
    ctx.synchronize()
    c = ctr.view(C4_RAYS, 2).to(torch.int64).sum(dim=0).cpu().numpy()
    del ctr
    ctx.set_option("time_kernels", 2)
    ctx.kernel_time()
    t0 = time.perf_counter()
    for _ in range(launches):
        scene.trace_raw(d_rays.data_ptr(), C4_RAYS, hits.data_ptr())
    ctx.synchronize()
    wall = time.perf_counter() - t0
    ms, n = ctx.kernel_time()
    ctx.set_option("time_kernels", 1)
    assert n == launches
    kernel_ms = ms / launches
    n_hit = int((hits.view(torch.int32).view(C4_RAYS, 4)[:, 1] != -1).sum().item())
    rec = {"workload": f"config 4: {C4_RAYS} bench_mesh rays (seed {C4_RAY_SEED}) vs {n_leaves}-triangle BVH "
                       f"({n_nodes} nodes, depth {depth}, built on the GPU), closest-hit ids",
           "value": round(C4_RAYS * launches / wall / 1e6, 1), "unit": "Mrays/s", "launches": launches,
           "ms_per_launch": round(wall / launches * 1e3, 3), "rays_hit": n_hit, "setup_s": round(setup_s, 2),
           "roofline": roofline_record("k_trace_phased", "config4-64Mrays-10Mtris", kernel_ms, C4_RAYS,
                                       c[0] / C4_RAYS, c[1] / C4_RAYS, 48.0, True, (n_nodes, n_leaves))}
    scene.close()
    mesh.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each; the median is reported")
    ap.add_argument("--traversal", type=int, default=-1, help="0 short stack, 1 threaded, 2 phased threaded, -1 library default")
    ap.add_argument("--opt", action="append", default=[], help="ctx option key=value (e.g. tri_min=16)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-c4", action="store_true", help="skip the configuration 4 record")
    ap.add_argument("--no-c3", action="store_true", help="skip the configuration 3 record")
    ap.add_argument("--no-c5", action="store_true", help="skip the configuration 5 record")
    ap.add_argument("--no-configs", action="store_true", help="config 2 only: skip the records of configurations 3, 4 and 5")
    ap.add_argument("--gather-timeout", type=float, default=90.0,
                    help="with a gather: seconds the first warm-up step's gather may take before the run gives up")
    ap.add_argument("--grid", type=str, default=None, help="NXxNY quads (default 1000x500)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="frames in flight per GPU: that many contexts (one stream each) take the frames in turn")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the gather path even with one rank (rehearses the multi-GPU step on a 1-GPU box)")
    ap.add_argument("--loopback", action="store_true",
                    help="with --force-dist on one rank: the gather goes through RCCL (the rank sends to itself) instead "
                         "of device copies, so RCCL's kernels run next to the traversal as they do with several ranks")
    ap.add_argument("--hit-buffers", type=int, default=2,
                    help="with a gather: hit buffers (and receive slots) per context, so a gather may lag the traversal by "
                         "that many frames before it holds it up")
    ap.add_argument("--root-in-place", choices=["auto", "on", "off"], default="auto",
                    help="rank 0 traces its frames directly into the gathered buffer (auto: when N>1 and the gather is native)")
    ap.add_argument("--fanin", type=int, default=1,
                    help="with --force-dist --loopback: frames the rank receives from itself per step (8 = the root's load "
                         "in an 8-GPU run)")
    ap.add_argument("--gather", choices=["native", "torch"], default="native",
                    help="native: rtg_gather_tiles behind the C ABI (RCCL send/recv); torch: torch.distributed.gather")
    args = ap.parse_args()
    if args.no_configs:
        args.no_c3 = args.no_c4 = args.no_c5 = True

    import torch
    import torch.distributed as dist
    from raytracer_amd import rtg, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    torch.cuda.set_device(local_rank)
    json_out = sys.stdout
    if use_dist:
        # RCCL prints a version banner on stdout when it creates a communicator: the one JSON line goes to the real
        # stdout, everything else the process writes there is sent to stderr
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    if args.gpus != world:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using {world}", file=sys.stderr)
    dev = torch.device("cuda", local_rank)

    nx, ny = GRID if args.grid is None else tuple(int(v) for v in args.grid.lower().split("x"))
    t0 = time.time()
    tris = rtg.heightfield(nx, ny, SEED, LOCAL_SCALE)
    bvh = rtg.HostBvh(tris)                       # host-side BvhForest::subdivide + Bvh::new
    build_s = time.time() - t0
    # A persistent launch ends with a tail in which its last rays finish with the GPU half empty (a ray is a chain of
    # ~90 dependent node fetches), so consecutive frames are given to `inflight` contexts in turn, each with its own
    # stream: the next frame's blocks take over the block slots the previous frame's tail frees.  The contexts
    # share one uploaded tree.
    inflight = max(1, args.inflight)
    ctxs, streams, scenes = [], [], []
    mesh = None
    for s_ in range(inflight):
        c = rtg.Context(local_rank)
        if args.traversal >= 0:
            c.set_option("traversal", args.traversal)
        if use_dist:
            # The persistent kernel fills every wave slot of the GPU for the whole frame (five waves per SIMD = 1280
            # four-wave groups), so a gather kernel on the second stream could only start when a frame drains.
            # Leaving 32 groups free lets RCCL's send/recv blocks run next to the traversal of the following frame.
            c.set_option("max_blocks", 1248)
        for kv in args.opt:
            k, v = kv.split("=")
            c.set_option(k, int(v))
        st = torch.cuda.Stream(dev)               # a real (non-null) stream: kernels, events and RCCL share it
        c.set_stream(st.cuda_stream)
        if mesh is None:
            mesh = c.mesh(bvh)
            obj = rtg.mesh_object(mesh, rtg.material((0.8, 0.7, 0.6)), rtg.transform(1.0 / LOCAL_SCALE))
        ctxs.append(c)
        streams.append(st)
        scenes.append(c.scene([obj]))
    ctx, stream, scene = ctxs[0], streams[0], scenes[0]
    torch.cuda.set_stream(stream)

    n_rays = WIDTH * HEIGHT
    tile = (0, 0, WIDTH, HEIGHT)
    frame_bytes = n_rays * 16
    n_buf = max(2, args.hit_buffers) if use_dist else 1      # hit buffers per context: a gather may lag that many frames
    hits_of = [[torch.empty(frame_bytes, dtype=torch.uint8, device=dev) for _ in range(n_buf)] for _ in range(inflight)]
    hits = hits_of[0][0]
    sizes = [hits.numel()] * world
    gather_kind = None
    if use_dist:
        comm_stream = torch.cuda.Stream(dev)          # gathers run here, traversal on the contexts' streams
        rendered = [[torch.cuda.Event() for _ in range(n_buf)] for _ in range(inflight)]
        gathered = [[torch.cuda.Event() for _ in range(n_buf)] for _ in range(inflight)]
        comm = None
        if args.gather == "native":
            # the library's own communicator: rank 0 makes the id, the others get it over the process group
            try:
                uid = torch.zeros(rtg.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
                if rank == 0 and world > 1:
                    uid.copy_(torch.frombuffer(bytearray(rtg.comm_unique_id()), dtype=torch.uint8))
                if world > 1:
                    dist.broadcast(uid, src=0)
                if world == 1 and args.loopback:      # rehearsal: one rank, but through RCCL (send / recv to itself)
                    ctx.set_option("comm_loopback", 2)
                    comm = rtg.Comm(ctx, 1, 0, rtg.comm_unique_id())
                else:
                    comm = rtg.Comm(ctx, world, rank, bytes(uid.cpu().numpy().tobytes()) if world > 1 else None)
                gather_kind = "rtg_gather_tiles (RCCL send/recv behind the C ABI)" + (", one rank sending to itself" if world == 1 and args.loopback else "")
            except Exception as e:                    # noqa: BLE001 -- said on stderr; the run ends below
                print(f"rank {rank}: the native communicator could not be created: {e}", file=sys.stderr, flush=True)
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
            if world > 1:                             # every rank learns of a failure anywhere and leaves together
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                # measuring another gather than the one asked for would change what the line means: no fallback
                print(f"rank {rank}: no native gather on every rank; ask for the torch path with --gather torch",
                      file=sys.stderr, flush=True)
                if comm is not None:
                    comm.close()
                if world > 1:
                    dist.destroy_process_group()
                sys.exit(3)
        else:
            gather_kind = "torch.distributed.gather"
        pieces = shard.frame_pieces(world, frame_bytes)
        fanin = 1
        if world == 1 and args.loopback and comm is not None and args.fanin > 1:
            # rehearsal of the root's side of a gather from `fanin` ranks: the one rank sends its frame that many times
            # to itself, into as many slots, through RCCL -- the receive traffic and the RCCL kernels of an N-GPU run
            # next to the root's own traversal (the links themselves are what a single GPU cannot rehearse)
            fanin = args.fanin
            pieces = [(0, 0, k * frame_bytes, frame_bytes) for k in range(fanin)]
            gather_kind += f", {fanin} frames received per step"
        if comm is not None:
            recv = [[torch.empty(max(world, fanin) * frame_bytes, dtype=torch.uint8, device=dev) if rank == 0 else None
                     for _ in range(n_buf)] for _ in range(inflight)]
        else:
            recv = [[[torch.empty_like(hits) for _ in range(world)] if rank == 0 else None for _ in range(n_buf)]
                    for _ in range(inflight)]
        in_place = comm is not None and rank == 0 and fanin == 1 and (
            args.root_in_place == "on" or (args.root_in_place == "auto" and world > 1))
        if in_place:
            # the root traces its own frames straight into its slot of the gathered buffer: the gather then has no
            # copy to make for them (the library skips a piece whose source and destination are the same memory)
            hits_of = [[recv[k][b][:frame_bytes] for b in range(n_buf)] for k in range(inflight)]
            hits = hits_of[0][0]
            gather_kind += ", the root's own frames traced in place"

    def step(i, ev=None, probe=None):
        """One frame.  `ev`: timing events around the traversal call; `probe` (with a gather): four more timing events
        -- the traversal stream before and after its wait for the hit buffer's previous gather, the gather's stream
        before and after the gather."""
        frame = i * world + rank
        k = i % inflight                              # the context that takes this frame
        b = (i // inflight) % n_buf if use_dist else 0
        st = streams[k]
        if use_dist:
            if probe is not None:
                probe[0].record(st)
            st.wait_event(gathered[k][b])             # the hit buffer is free once its previous gather has finished
            if probe is not None:
                probe[1].record(st)
        if ev is not None:
            ev[0].record(st)
        scenes[k].render_primary_raw(frame_view(rtg, frame), WIDTH, HEIGHT, tile, hits_of[k][b].data_ptr())
        if ev is not None:
            ev[1].record(st)
        if use_dist:
            rendered[k][b].record(st)
            comm_stream.wait_event(rendered[k][b])
            if probe is not None:
                probe[2].record(comm_stream)
            if comm is not None:                      # gather of the finished framebuffers to rank 0
                comm.gather(0, hits_of[k][b].data_ptr(), recv[k][b].data_ptr() if rank == 0 else None, pieces,
                            stream_handle=comm_stream.cuda_stream)
            else:
                with torch.cuda.stream(comm_stream):
                    shard.gather_to_root(hits_of[k][b], sizes, dst=0, out=recv[k][b])
            if probe is not None:
                probe[3].record(comm_stream)
            gathered[k][b].record(comm_stream)

    for i in range(args.warmup):
        step(i)
        if use_dist and i == 0:
            # The first gather is where a transport that cannot be set up shows (the connections between the ranks
            # are made on first use): it gets a bounded wait, polled from the host, instead of the device-wide
            # synchronize below, which would never return.  What is pending is said before the process leaves.
            deadline = time.time() + args.gather_timeout
            while not gathered[0][0].query():
                if time.time() > deadline:
                    print(f"rank {rank}: the first gather ({gather_kind}; {len(pieces)} pieces of {frame_bytes} bytes, "
                          f"{world} ranks) has not completed after {args.gather_timeout:.0f} s; traversal of the frame "
                          f"{'finished' if rendered[0][0].query() else 'not finished either'}; giving up",
                          file=sys.stderr, flush=True)
                    os._exit(4)
                time.sleep(0.01)
    # one untimed window of the timed windows' size: clocks and caches settle (the first window otherwise reads
    # 2-4 % low whatever --warmup is)
    for i in range(args.steps):
        step(args.warmup + i)
    settle_steps = args.steps
    # The kernel's duration for the roofline: HIP events that the launch itself fills with its execution window
    # (the dispatch's own begin and end timestamps -- what rocprofv3 reports per dispatch).  Events recorded around the call
    # would also count the time a launch waits in its stream's queue while the previous frame still owns the GPU.
    timed_kernels = args.traversal in (-1, 2)
    n_windows = max(1, args.windows)
    window_s, window_kernel_ms, window_gather = [], [], []
    done = args.warmup + settle_steps
    for w_ in range(n_windows):
        if timed_kernels:
            torch.cuda.synchronize(dev)
            for c in ctxs:
                c.set_option("time_kernels", 2)
                c.kernel_time()
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        probes = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)] if use_dist else None
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(done + i, events[i], probes[i] if probes else None)
        torch.cuda.synchronize(dev)                   # every stream: every frame rendered and gathered
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        done += args.steps
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
        if timed_kernels:
            windows = [c.kernel_time() for c in ctxs]
            for c in ctxs:
                c.set_option("time_kernels", 1)
            assert sum(n for _, n in windows) == args.steps
            kernel_ms = sum(ms for ms, _ in windows) / args.steps
        window_s.append(elapsed)
        window_kernel_ms.append(kernel_ms)
        if use_dist:                                  # this rank's waits and gathers, per step
            window_gather.append((float(np.mean([p[0].elapsed_time(p[1]) for p in probes])),
                                  float(np.mean([p[2].elapsed_time(p[3]) for p in probes]))))
    order = np.argsort(window_s)
    mid = int(order[(n_windows - 1) // 2])            # the median window (the lower one of an even count)
    elapsed, kernel_ms = window_s[mid], window_kernel_ms[mid]
    gather_rec = None
    if use_dist:
        # every rank's wait for a free hit buffer and its gather's duration in the median window, collected on rank 0
        mine = torch.tensor(list(window_gather[mid]), dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)] if world > 1 else [mine]
        if world > 1:
            dist.all_gather(every, mine)
        waits = [float(t[0].item()) for t in every]
        durs = [float(t[1].item()) for t in every]
        ranks_info = comm.info() if comm is not None else (world, rank, None)
        sent = sum(p[3] for p in pieces if p[0] != 0)          # bytes that cross a link per step
        root_ms = durs[0]
        gather_rec = {
            "kind": gather_kind, "ranks": world, "rccl_ranks": ranks_info[2],
            "bytes_per_step_per_rank": frame_bytes, "bytes_into_root_per_step": sent,
            "gather_ms_per_step_by_rank": [round(v, 4) for v in durs],
            "wait_for_hit_buffer_ms_per_step_by_rank": [round(v, 4) for v in waits],
            "gbs_per_sending_rank": [None] + [round(frame_bytes / (v * 1e-3) / 1e9, 1) if v > 0 else None for v in durs[1:]],
            "gbs_into_root": round(sent / (root_ms * 1e-3) / 1e9, 1) if (root_ms > 0 and sent) else None,
            "link_budget": "one xGMI link per peer, about 153 GB/s both ways together, so about 77 GB/s one way (the "
                           f"platform's nominal figure, not measured here): a sending rank moves {frame_bytes / 1e6:.1f} MB "
                           "per step over its link to the root, the root takes (ranks - 1) such frames over as many links",
        }

    if rank == 0:
        # node visits and algorithmic bytes of one launch, from the kernel's own counting build on frame 0
        ctr = torch.empty(n_rays * 2, dtype=torch.int32, device=dev)
        scene.render_primary_raw(frame_view(rtg, 0), WIDTH, HEIGHT, tile, hits.data_ptr(), None, ctr.data_ptr())
        torch.cuda.synchronize(dev)
        c = ctr.view(n_rays, 2).to(torch.int64).sum(dim=0).cpu().numpy()
        del ctr
        nodes_per_ray, tris_per_ray = c[0] / n_rays, c[1] / n_rays
        mrays = args.steps * world * n_rays / elapsed / 1e6
        window_mrays = [round(args.steps * world * n_rays / s / 1e6, 1) for s in window_s]

        cpu = None
        if not args.no_cpu and world == 1:          # the CPU leg is reported for the single-GPU line only
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import orc
            cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
            obvh = orc.OracleBvh(flat=bvh.view)
            oscene = orc.OracleScene([orc.oracle_object(obj, obvh)])
            # bounded sample: whole frames of the sequence until about 1.5 s of wall time (~25 core-seconds)
            cpu_s, cpu_frames, c_hits = 0.0, 0, None
            while cpu_s < 1.5 and cpu_frames < 16:
                t0 = time.perf_counter()
                h, _, _ = oscene.render_primary(frame_view(rtg, cpu_frames), WIDTH, HEIGHT, threads=cores)
                cpu_s += time.perf_counter() - t0
                if cpu_frames == 0:
                    c_hits = h
                cpu_frames += 1
            scene.render_primary_raw(frame_view(rtg, 0), WIDTH, HEIGHT, tile, hits.data_ptr())
            torch.cuda.synchronize(dev)
            g = np.frombuffer(hits.cpu().numpy().tobytes(), dtype=rtg.HIT_DTYPE)
            t_diff = (g["t"].view(np.uint64) != c_hits["t"].view(np.uint64)) & ~(np.isnan(g["t"]) & np.isnan(c_hits["t"]))
            mism = int(((g["object"] != c_hits["object"]) | (g["leaf"] != c_hits["leaf"]) | t_diff).sum())
            cpu = {"value": round(cpu_frames * n_rays / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                   "sample": f"{cpu_frames} full {WIDTH}x{HEIGHT} frames (frames 0..{cpu_frames - 1}), "
                             f"{cpu_frames * n_rays} rays, {cpu_s:.2f} s wall on {cores} threads",
                   "hit_id_mismatches_vs_gpu": mism}

        kernel = KERNEL_OF_TRAVERSAL[args.traversal]
        default_run = args.traversal < 0 and not args.opt and args.grid is None
        roof = roofline_record(kernel, "config2-1920x1080", kernel_ms, n_rays, nodes_per_ray, tris_per_ray, 0.0, default_run,
                               (bvh.n_nodes, 2 * nx * ny))
        roof["concurrent_launches"] = inflight
        # the same visits against the whole timed region: what the GPU sustains with `inflight` launches overlapping
        roof["gather"]["achieved_timed_region"] = round(nodes_per_ray * n_rays * args.steps / elapsed / 1e9, 2)
        if roof["gather"]["peak"]:
            roof["gather"]["frac_timed_region"] = round(roof["gather"]["achieved_timed_region"] / roof["gather"]["peak"], 4)
        configs = None
        if world == 1 and default_run and not args.force_dist:
            torch.cuda.set_stream(stream)
            configs = {}
            if not args.no_c5:
                configs["c5"] = run_config5(rtg, torch, ctx, dev, scene)
            if not args.no_c3:
                configs["c3"] = run_config3(rtg, torch, ctx, dev)
            if not args.no_c4:
                configs["c4"] = run_config4(rtg, torch, ctx, dev)
            configs = configs or None
        out = {
            "metric": "Mrays/s, 1M-tri BVH scene", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "windows": n_windows, "settle_steps": settle_steps, "windows_mrays": window_mrays,
            "spread": round((max(window_mrays) - min(window_mrays)) / float(np.median(window_mrays)), 4),
            "config": {"workload": f"config 2: {2 * nx * ny}-triangle terrain BVH ({bvh.n_nodes} nodes, depth {bvh.depth}), "
                                   f"{WIDTH}x{HEIGHT} primary rays per step per GPU, closest-hit ids",
                       "traversal": "library default" if args.traversal < 0 else ["short-stack", "threaded", "phased-threaded"][args.traversal],
                       "parallelism": f"frames sharded over {world} GPU(s), BVH replicated"
                                      + (f", gather to rank 0 by {gather_kind}, overlapped with the next frame" if use_dist else "")
                                      + f", {inflight} frame(s) in flight per GPU",
                       "host_bvh_build_s": round(build_s, 2)},
            "roofline": roof,
            "cpu_baseline": cpu,
            "configs": configs,
        }
        if gather_rec is not None:
            out["gather"] = gather_rec
        print(json.dumps(out), file=json_out, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
