#!/usr/bin/env python3
"""bench.py -- Mrays/s of the GPU ray-cast path on the 1M-triangle BVH scene.

A step is one pass of the hot path over one batch: a full 1920x1080 primary-ray frame (ray generation +
BVH traversal + ray/triangle intersection + closest hit -> 16-byte hit records) against the 1 000 000-triangle
terrain, inputs (scene, BVH) resident in HBM, outputs written to HBM.  With N > 1 every rank renders its own
frame of the sequence (frame f -> rank f mod N, camera moved per frame; BVH replicated) and the hit buffers are
gathered to rank 0 with RCCL inside the timed region, on a second stream so the gather of frame i overlaps the
traversal of frame i+1 (two hit buffers per rank): weak scaling.  Every GPU keeps `--inflight` frames in flight
(default 2: two contexts with a stream each take the frames in turn, sharing one uploaded tree), because the tail of
a persistent launch -- its last rays finishing on a half-empty GPU -- is best hidden under the next frame's start.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (k_trace_phased, the persistent traversal
kernel behind rtg_render_primary) against HBM peak using ALGORITHMIC bytes (nodes visited x 64 B + triangles tested
x 72 B + 16 B hit record per ray, counted by the kernel's counting build on the same frame) over the kernel's average
launch duration from HIP events on the launching stream -- with two launches in flight a launch spans the time it
shares the GPU with its neighbour, so `roofline.concurrent_launches` states how many overlap; `roofline.traffic` is
the HBM traffic per launch from the rocprofv3 counter passes committed under profiles/ (2 x FETCH_SIZE + WRITE_SIZE),
null when no committed pass matches the workload; `cpu_baseline` is the oracle (C++ restatement of the reference
renderer, kind "port") timed on this host's cores on whole frames of the same sequence.
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


def frame_view(rtg, frame):
    """Camera of frame `frame` of the sequence: the default view dollying slowly sideways."""
    return rtg.make_view(orig=(0.002 * frame, 0.0, 1.0))


KERNEL_OF_TRAVERSAL = {-1: "k_trace_phased", 0: "k_render_primary", 1: "k_render_primary", 2: "k_trace_phased"}


def committed_traffic(workload_key, kernel):
    """HBM bytes per launch of `kernel` from the newest committed counter summary for this workload, or None."""
    best = None
    prof = os.path.join(ROOT, "profiles")
    if not os.path.isdir(prof):
        return None
    for d in sorted(os.listdir(prof)):
        path = os.path.join(prof, d, "traffic.json")
        if os.path.isfile(path):
            with open(path) as f:
                t = json.load(f)
            if t.get("workload") == workload_key and t.get("kernel") == kernel:
                best = t
    return None if best is None else float(best["hbm_bytes_per_launch"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--traversal", type=int, default=-1, help="0 short stack, 1 threaded, 2 phased threaded, -1 library default")
    ap.add_argument("--opt", action="append", default=[], help="ctx option key=value (e.g. tri_min=16)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--grid", type=str, default=None, help="NXxNY quads (default 1000x500)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="frames in flight per GPU: that many contexts (one stream each) take the frames in turn")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the gather path even with one rank (rehearses the multi-GPU step on a 1-GPU box)")
    args = ap.parse_args()

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
            # The persistent kernel fills every block slot of the GPU for the whole frame (128 VGPRs x 4 waves per
            # SIMD), so a gather kernel on the second stream could only start when a frame drains.  Leaving 32 of
            # the 1024 slots free lets RCCL's send/recv blocks run next to the traversal of the following frame.
            c.set_option("max_blocks", 992)
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
    n_buf = 2 if use_dist else 1
    hits_of = [[torch.empty(n_rays * 16, dtype=torch.uint8, device=dev) for _ in range(n_buf)] for _ in range(inflight)]
    hits = hits_of[0][0]
    sizes = [hits.numel()] * world
    if use_dist:
        comm = torch.cuda.Stream(dev)                 # gathers run here, traversal on the contexts' streams
        recv = [[[torch.empty_like(hits) for _ in range(world)] if rank == 0 else None for _ in range(2)]
                for _ in range(inflight)]
        rendered = [[torch.cuda.Event() for _ in range(2)] for _ in range(inflight)]
        gathered = [[torch.cuda.Event() for _ in range(2)] for _ in range(inflight)]

    def step(i, ev=None):
        frame = i * world + rank
        k = i % inflight                              # the context that takes this frame
        b = (i // inflight) & 1 if use_dist else 0
        st = streams[k]
        if use_dist:
            st.wait_event(gathered[k][b])             # the hit buffer is free once its previous gather has finished
        if ev is not None:
            ev[0].record(st)
        scenes[k].render_primary_raw(frame_view(rtg, frame), WIDTH, HEIGHT, tile, hits_of[k][b].data_ptr())
        if ev is not None:
            ev[1].record(st)
        if use_dist:
            rendered[k][b].record(st)
            comm.wait_event(rendered[k][b])
            with torch.cuda.stream(comm):             # RCCL gather of the finished framebuffers to rank 0
                shard.gather_to_root(hits_of[k][b], sizes, dst=0, out=recv[k][b])
                gathered[k][b].record(comm)

    for i in range(args.warmup):
        step(i)
    # The kernel's duration for the roofline: HIP events that the launch itself fills with its execution window
    # (the dispatch's own begin and end timestamps -- what rocprofv3 reports per dispatch).  Events recorded around the call
    # would also count the time a launch waits in its stream's queue while the previous frame still owns the GPU.
    timed_kernels = args.traversal in (-1, 2)
    if timed_kernels:
        torch.cuda.synchronize(dev)
        for c in ctxs:
            c.set_option("time_kernels", 2)
            c.kernel_time()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, events[i])
    torch.cuda.synchronize(dev)                       # both streams: every frame rendered and gathered
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
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

    if rank == 0:
        # algorithmic bytes of one launch, from the kernel's own counting build on frame 0
        ctr = torch.empty(n_rays * 2, dtype=torch.int32, device=dev)
        scene.render_primary_raw(frame_view(rtg, 0), WIDTH, HEIGHT, tile, hits.data_ptr(), None, ctr.data_ptr())
        torch.cuda.synchronize(dev)
        c = ctr.view(n_rays, 2).to(torch.int64).sum(dim=0).cpu().numpy()
        nodes_per_ray, tris_per_ray = c[0] / n_rays, c[1] / n_rays
        bytes_per_ray = nodes_per_ray * 64.0 + tris_per_ray * 72.0 + 0.0 + 16.0
        alg_bytes = bytes_per_ray * n_rays
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        mrays = args.steps * world * n_rays / elapsed / 1e6

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
        traffic = committed_traffic("config2-1920x1080", kernel) if default_run else None
        out = {
            "metric": "Mrays/s, 1M-tri BVH scene", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"config 2: {2 * nx * ny}-triangle terrain BVH ({bvh.n_nodes} nodes, depth {bvh.depth}), "
                                   f"{WIDTH}x{HEIGHT} primary rays per step per GPU, closest-hit ids",
                       "traversal": "library default" if args.traversal < 0 else ["short-stack", "threaded", "phased-threaded"][args.traversal],
                       "parallelism": f"frames sharded over {world} GPU(s), BVH replicated"
                                      + (", RCCL gather to rank 0 overlapped with the next frame" if use_dist else "")
                                      + f", {inflight} frame(s) in flight per GPU",
                       "host_bvh_build_s": round(build_s, 2)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None if traffic is None else round(traffic / 1e9, 4), "traffic_unit": "GB per launch",
                         "kernel": kernel, "concurrent_launches": inflight,
                         "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_ray": round(bytes_per_ray, 1),
                         "nodes_per_ray": round(float(nodes_per_ray), 2), "tris_per_ray": round(float(tris_per_ray), 2)},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), file=json_out, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
