#!/usr/bin/env python3
"""bench.py — Mrays/s and frame time of the render hot path on MI355X.

A step = one frame of BASELINE.json's headline configuration (configs[1]: 1920x1080, 8 spheres +
1 plane, 3 lights, depth 3, no AA; synthetic scene = the reference's default scene + 2 spheres).
With N > 1 ranks (one process per GPU, launched by torch.distributed.run — or by this script itself
when it is started plainly with --gpus N) the SAME frame is cut into N column slabs, each rank renders
its slab, and the uint8 frames are assembled on rank 0 by RCCL gathers (the slabs of several consecutive
frames per gather) — total work is fixed, so `scaling` is "strong".
Frames are queued round-robin on a few streams so that consecutive frames overlap.

Inputs (scene, camera) are resident on the device before the timed region; outputs stay in HBM.
Before the warm-up the device is pre-heated for a fixed WALL TIME with real frames (`preheat_ms`): the
engine clock needs a few hundred milliseconds of load to reach its steady state, and `--steps 20` on a
cold device times 2.6 ms of it.  `--steps` / `--warmup` are honoured exactly as passed.
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time

# HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, counting the
# default stream and every library-internal one; two of the bench's three render streams can land on the same
# queue, where their frames serialise (measured: 0.112 instead of 0.108 ms per frame).  Ask for 8 queues before
# the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
ROUND = "r03"           # profiles/traffic_<ROUND>[_<workload>].json, profiles/valu_<ROUND>[_<workload>].json (tools/profile_round.py --stamp)
PRICES = "r03_valu_prices.json"   # cycles per wave-instruction and class, measured by tools/valu_prices.hip on an MI355X
CLOCK_HZ = 2.4e9        # peak engine clock: a kernel cannot run faster than its priced instructions take at this clock
SIMDS = 1024            # 256 CUs x 4
# which measured class prices which rocprofv3 op-mix counter (SQ_INSTS_VALU_<KEY>)
PRICE_CLASS = {"add_f64": "v_add_f64", "mul_f64": "v_mul_f64", "fma_f64": "v_fma_f64", "trans_f64": "v_rsq_f64",
               "add_f32": "v_add_f32", "mul_f32": "v_mul_f32", "fma_f32": "v_fma_f32", "trans_f32": "v_rcp_f32",
               "cvt": "v_cvt_f32_f64", "int32": "v_and_b32", "int64": "v_lshlrev_b64"}
UNCATEGORISED_FLOOR = "v_mov_b32"                   # the cheapest VALU class measured: prices the lower bound
UNCATEGORISED_TYPICAL = "v_cmp_lt_f64_e64->sgpr"    # compares, selects, readlanes, 64-bit moves all measure 4.1-4.2 cycles


def cpu_baseline(wl, rays_per_frame, min_wall_s=2.5):
    """The CPU oracle (a C restatement of the reference, kind="port") timed on this box's host cores
    on whole frames of the same workload, until at least `min_wall_s` of wall time has been spent."""
    from oracle import oracle as orc
    orc.build()
    cam = wl["camera"]
    threads = orc.max_threads()
    frames, t0 = 0, time.perf_counter()
    while True:
        orc.render(wl["w"], wl["h"], cam.position, cam.rotation, wl["spheres"], wl["lights"], wl["planes"],
                   wl["amb"], wl["lamb"], wl["refl"], wl["depth"], wl["aa"], raygen=cam.raygen(), want=("u8",),
                   nthreads=threads, spp=wl["spp"], seed=wl["seed"])
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= min_wall_s:
            break
    return {"value": round(rays_per_frame * frames / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{frames} full {wl['w']}x{wl['h']} frame(s) of the same workload, {dt:.2f} s wall, "
                      f"{dt * threads:.1f} core-seconds, OpenMP over columns",
            "frame_ms": round(dt / frames * 1e3, 2)}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (before this
    process has touched the GPU) and pass their output and exit code through."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def issue_bound(v, prices, waves_per_simd):
    """Prices a kernel's measured VALU op mix (wave-instructions per launch and counter class) with the measured cycles
    per wave-instruction at the kernel's occupancy.  The instructions the counters do not classify (compares, selects,
    moves, readlanes, DPP ...) are priced at the cheapest measured class for the BOUND (a launch cannot take less) and
    at the price of a compare/select for the ESTIMATE."""
    col = "w7" if waves_per_simd >= 6 else "w4"
    cyc = {c["class"]: c[col]["cycles_simd_span"] for c in prices["classes"]}
    mix = v["op_mix_wave_instructions"]
    total = v["valu_wave_instructions_per_launch"]
    cycles, counted, table = 0.0, 0, {}
    for key, cls in PRICE_CLASS.items():
        n = mix.get(key, 0)
        counted += n
        cycles += n * cyc[cls]
        table[key] = {"wave_instructions": n, "cycles_each": cyc[cls], "priced_as": cls}
    unc = max(0, total - counted)
    table["uncategorised"] = {"wave_instructions": unc, "cycles_each_bound": cyc[UNCATEGORISED_FLOOR], "priced_as_bound": UNCATEGORISED_FLOOR,
                              "cycles_each_estimate": cyc[UNCATEGORISED_TYPICAL], "priced_as_estimate": UNCATEGORISED_TYPICAL}
    lo = (cycles + unc * cyc[UNCATEGORISED_FLOOR]) / (SIMDS * CLOCK_HZ) * 1e3
    est = (cycles + unc * cyc[UNCATEGORISED_TYPICAL]) / (SIMDS * CLOCK_HZ) * 1e3
    return lo, est, table, col


def stamped(path, so_sha):
    """A profile-derived constant file applies only to the library build it was measured on."""
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    return d if d.get("so_sha256") == so_sha else None


class HostStagedGroup:
    """REHEARSAL ONLY (--rehearse-gloo): the torch.distributed calls this file and SequencePipeline make, on the gloo backend
    with every device tensor staged through host memory.  RCCL refuses two ranks on one device, so this is the only way to run
    the N > 1 control flow (slab balancing, batched gathers with a rotating root, per-rank statistics) on a one-GPU box: N
    processes share cuda:0.  The frames are checked like any other run's; the timings mean nothing."""

    class _Done:
        def wait(self):
            return True

    def __init__(self, dist, torch):
        self.d, self.torch, self.ReduceOp = dist, torch, dist.ReduceOp

    def get_rank(self):
        return self.d.get_rank()

    def get_world_size(self):
        return self.d.get_world_size()

    def barrier(self):
        self.d.barrier()

    def destroy_process_group(self):
        self.d.destroy_process_group()

    def all_reduce(self, t, op=None):
        c = t.cpu()
        self.d.all_reduce(c, op=op if op is not None else self.d.ReduceOp.SUM)
        t.copy_(c)

    def broadcast(self, t, src=0):
        c = t.cpu()
        self.d.broadcast(c, src=src)
        t.copy_(c)

    def all_gather(self, outs, t):
        cs = [self.torch.empty(o.shape, dtype=o.dtype) for o in outs]
        self.d.all_gather(cs, t.cpu())
        for o, c in zip(outs, cs):
            o.copy_(c)

    def gather(self, t, recv, dst=0, async_op=False):
        c = t.cpu()                                     # (on the current stream, behind whatever it waits for; then the host waits)
        rc = [self.torch.empty(c.shape, dtype=c.dtype) for _ in recv] if recv is not None else None
        self.d.gather(c, rc, dst=dst)
        if recv is not None:
            for o, x in zip(recv, rc):
                o.copy_(x)
        return self._Done()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default=None, help="one of python_ray_tracer_amd.workloads.CONFIGS")
    ap.add_argument("--streams", type=int, default=None, help="streams the launches are queued on round-robin; default 1 on one GPU (a launch "
                    "renders --frames-per-launch frames in one grid: they overlap inside the launch) and 3 with a gather")
    ap.add_argument("--frames-per-gather", type=int, default=8, help="N > 1: frames whose slabs travel to rank 0 in one gather")
    ap.add_argument("--frames-per-launch", type=int, default=16, help="frames one kernel launch renders (rt_render_sequence); with a gather: = --frames-per-gather; 0 = one Python call and one launch per frame, as in round 2")
    ap.add_argument("--gather", choices=("u8", "f32"), default="u8", help="N > 1: assemble the uint8 frames only, or the float32 pre-clip planes as well (a second gather per batch)")
    ap.add_argument("--no-dynamic", action="store_true", help="skip the moving-camera pass behind the timed region (dynamic)")
    ap.add_argument("--force-gather", action="store_true", help="run the N > 1 exchange structure on one GPU (world-size-1 RCCL group)")
    ap.add_argument("--preheat-ms", type=float, default=300.0, help="wall time of real frames rendered before the warm-up (clock ramp)")
    ap.add_argument("--no-step-events", action="store_true", help="do not record one HIP event per step (drops the per-step statistics)")
    ap.add_argument("--no-serial", action="store_true", help="skip the one-stream pass behind the timed region (roofline.serial)")
    ap.add_argument("--no-host-path", action="store_true", help="skip timing the host-buffer entry point rt_render (host_path_ms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="extra rt_params.flags for every launch (e.g. 4 = RT_FLAG_NO_FEEDBACK; profiling variants)")
    ap.add_argument("--rehearse-gloo", action="store_true", help="N > 1 on ONE GPU: all ranks share cuda:0, collectives on gloo through host copies "
                    "(HostStagedGroup): checks the control flow and the assembled frames, its timings mean nothing")
    ap.add_argument("--gather-root", choices=("rotate", "fixed"), default="rotate",
                    help="N > 1: assemble batch b on rank b %% N (every rank's links carry a share of the exchange) or always on rank 0")
    ap.add_argument("--balance-rounds", type=int, default=4, help="N > 1: rounds of measured-time slab balancing before the run (0 = equal-width slabs)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1 and "RANK" not in os.environ:
            raise SystemExit(spawn_ranks(a.gpus))
        a.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    import python_ray_tracer_amd as pkg
    from python_ray_tracer_amd import workloads, _lib
    from python_ray_tracer_amd.distributed import slab_bounds, SequencePipeline

    if a.rehearse_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if a.rehearse_gloo:
            dist.init_process_group("gloo")
            dist = HostStagedGroup(dist, torch)
        else:
            dist.init_process_group("nccl", device_id=dev)

    name = a.workload or workloads.HEADLINE
    wl = workloads.build(name)
    w, h, cam = wl["w"], wl["h"], wl["camera"]
    r = pkg.Renderer(local_rank)
    r.set_scene(wl["spheres"], wl["lights"], wl["planes"])
    r.set_camera(cam.position, cam.rotation)
    r.set_raygen(w, h, *cam.raygen())
    params = r.params(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], wl["aa"], spp=wl["spp"], seed=wl["seed"], flags=a.flags)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Column slabs: equal widths to start with; with N > 1 the boundaries are then moved until the ranks' MEASURED slab
    # times agree (sky columns are cheap, the sphere cluster is not: equal widths leave the slowest rank 23 % above
    # the mean at N = 8, and the frame is as slow as its slowest slab).  Setup, like the uploads: not timed.
    if a.streams is None:
        a.streams = 3 if (world > 1 or a.force_gather) else 1
    bounds = [slab_bounds(w, world, q) for q in range(world)]
    balance = None
    if world > 1 and a.balance_rounds > 0:
        bounds, balance = balance_slabs(r, params, w, h, world, rank, dev, dist, torch, max(1, a.streams), a.balance_rounds)
    x0, x1 = bounds[rank]
    ws = x1 - x0
    # python_ray_tracer_amd.distributed.SequencePipeline does the choreography (tests/test_distributed_cpu.py runs the
    # same class on gloo): frames are queued round-robin on `--streams` torch-owned, non-default streams (default 3),
    # each frame in flight with its own output buffers, so that one frame's last workgroups overlap the next
    # frame's first — a single in-order stream cannot do that; every frame is rendered in full, `--streams 1` is
    # the strictly serial variant.  With N > 1 the uint8 slabs of `--frames-per-gather` consecutive frames travel
    # to rank 0 in ONE gather (a collective costs tens of microseconds however small it is; a 240-column slab
    # renders in less), issued on a separate stream, two exchanges in flight: batch i is gathered and assembled
    # while batch i+1 renders.
    use_gather = world > 1 or a.force_gather
    NS = max(1, a.streams)
    if a.force_gather and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    batched = a.frames_per_launch > 0
    F = max(1, a.frames_per_gather) if use_gather else max(1, a.frames_per_launch)
    frame = frame32 = None

    def on_frames(first, frames, count, frames32=None):  # the batch's root: a batch of assembled (3,w,h) frames
        nonlocal frame, frame32
        frame = frames[count - 1]
        frame32 = frames32[count - 1] if frames32 is not None else None
    pipe = SequencePipeline(w, h, ws, dev, dist if use_gather else None, dst=0, streams=NS, frames_per_gather=F,
                            want_f32=True, on_frames=on_frames, bounds=bounds if use_gather else None,
                            rotate_root=(a.gather_root == "rotate" and world > 1), frames_per_launch=F,
                            gather_f32=(a.gather == "f32"))
    PS = pipe.plane_stride                              # slabs are stored padded to the widest rank's width
    assert all(pipe.stream_handle(i) for i in range(NS)), "expected non-default stream handles"

    ptrs = {}                                           # tensor view -> device address, looked up once per buffer

    def launch(u8, f32, stream):
        k = id(u8)
        if k not in ptrs:
            ptrs[k] = (u8.data_ptr(), f32.data_ptr())
        r.render_device(params, x0, x1, ptrs[k][0], ptrs[k][1], PS, stream)

    # A batch of F frames = ONE call of rt_render_sequence = one kernel launch (the frames share a grid): the host pays one
    # Python call and one launch per F frames.  The whole-batch tensors live as long as the pipeline, so their id() is a
    # safe key; slices (a partly filled batch) are looked up directly.
    batch_ptrs = {id(t): (t.data_ptr(), f.data_ptr()) for t, f in zip(pipe.u8, pipe.f32)}
    torch_stream = {hnd: st for hnd, st in zip(pipe.handles, pipe.streams)}
    event_pool = []                                     # events whose HIP objects exist already (torch creates them at the first record)

    def launch_seq(u8b, f32b, nf, stream):
        p8, p32 = batch_ptrs.get(id(u8b)) or (u8b.data_ptr(), f32b.data_ptr())
        r.render_sequence(params, x0, x1, nf, p8, p32, PS, 3 * PS, None, (stream,), nf)

    def submit(nframes, events=None):
        """Queue nframes frames; with `events`, one HIP event behind every launch, on the stream it was launched on."""
        if not batched:
            for _ in range(nframes):
                i = pipe.n
                pipe.submit(launch)
                if events is not None:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(pipe.streams[i % NS])
                    events.append((ev, i % NS, 1))
            return
        if events is None:
            pipe.submit_frames(launch_seq, nframes)
            return
        left = nframes
        while left > 0:                                 # batch by batch, so that every launch gets its event
            room = F - pipe.n % F
            for si, nf in pipe.submit_frames(launch_seq, min(left, room)):
                ev = event_pool.pop() if event_pool else torch.cuda.Event(enable_timing=True)
                ev.record(pipe.streams[si])
                events.append((ev, si, nf))
            left -= min(left, room)

    for _ in range(2):                                  # setup, like the uploads above: the first two launches of a geometry
        launch(pipe.u8v[0][0], pipe.f32v[0][0], pipe.stream_handle(0))
    torch.cuda.synchronize()                            # build the cull tables, measure tile costs, build the dispatch order

    # -- pre-heat: real frames for a fixed wall time (every rank the same number of frames: with N > 1 the frames
    # carry collectives).  A short calibration burst prices a frame, the count follows from it.
    t_pre = time.perf_counter()
    preheat_frames = 0
    if a.preheat_ms > 0:
        cal = 4 * F * NS
        tc = time.perf_counter()
        submit(cal)
        pipe.drain()
        per = (time.perf_counter() - tc) / cal
        n = int(max(0.0, a.preheat_ms * 1e-3 - (time.perf_counter() - t_pre)) / max(per, 1e-6))
        cnt = torch.tensor([n], dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
        n = min(int(cnt[0]), 2_000_000)
        n = -(-n // F) * F                              # whole launches (a kernel trace of the run then holds launches of one size)
        submit(n)
        pipe.drain()
        preheat_frames = cal + n
    preheat_ms = (time.perf_counter() - t_pre) * 1e3

    submit(a.warmup)
    pipe.drain()
    # One HIP event per launch stream in front of the timed region and one behind every LAUNCH, on the stream the
    # launch was made on (torch.cuda.Event on a stream object records on THAT stream, not on torch's current one).
    # Launches of one stream run back to back, so the gap between consecutive events of a stream is the duration of one
    # launch as rocprofv3's kernel trace sees it (a launch renders `frames_per_launch` frames); the gaps between completions
    # over all streams give the frame periods (over windows of completions, below).  NS launches are in flight at a time.
    step_events = not a.no_step_events
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(NS)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(NS)]
    evs = [] if step_events else None
    if step_events and batched:                         # (creating a HIP event costs microseconds of the 2 ms a --steps 20 run is timed over)
        event_pool.extend(torch.cuda.Event(enable_timing=True) for _ in range(min(4096, a.steps // F + 2)))
    for ev in ev0 + ev1 + event_pool:
        ev.record(pipe.streams[0])
    fence()
    t0 = time.perf_counter()
    for s_ in range(NS):
        ev0[s_].record(pipe.streams[s_])
    submit(a.steps, evs)
    t_submitted = time.perf_counter()
    for s_ in range(NS):
        ev1[s_].record(pipe.streams[s_])
    if world > 1 or use_gather:
        pipe.drain()                                    # every one of the K frames is assembled on its root
    fence()
    dt = time.perf_counter() - t0
    gpu_span_ms = max(ev0[s_].elapsed_time(ev1[s_]) for s_ in range(NS))   # first launch's start to the last one's end, device clock
    per_launch, periods, launch_frames = [], [], []
    kernel_ms, frames_per_launch_avg = 0.0, float(F if batched else 1)
    if step_events and evs:
        last = {s_: ev0[s_] for s_ in range(NS)}
        done = []
        for ev, si, nf in evs:
            per_launch.append(last[si].elapsed_time(ev))
            launch_frames.append(nf)
            last[si] = ev
            done.append((ev0[0].elapsed_time(ev), nf))   # completion time of the launch on a common clock
        kernel_ms = sum(per_launch) / len(per_launch)
        frames_per_launch_avg = sum(launch_frames) / len(launch_frames)
        # NS launches are in flight together and complete in clumps, so the period of a frame is taken over windows of
        # consecutive completions that hold at least 32 frames (all of them under --steps 20):
        # (t[i] - t[j]) / (frames completed after j up to i).  (Round 2 used 2 NS frames: a clump of completions could then
        # read as a period below the kernel's own issue bound.)
        done.sort()
        need = min(max(16 * NS, 32), max(1, sum(nf for _, nf in done[1:])))
        j, acc = 0, 0
        for i in range(1, len(done) if len(done) >= 4 * NS else 0):      # (a handful of launches complete together: no statistic)
            acc += done[i][1]
            while acc - done[j + 1][1] >= need and j + 1 < i:
                j += 1
                acc -= done[j][1]
            if acc >= need:
                periods.append((done[i][0] - done[j][0]) / acc)
    else:
        launches = [0] * NS
        pos = 0
        spans = [ev0[s_].elapsed_time(ev1[s_]) for s_ in range(NS)]
        nl = max(1, -(-a.steps // (F if batched else 1)))
        kernel_ms = sum(spans) / max(len(spans), 1) / max(1.0, nl / NS)
    if not use_gather:
        frame = pipe.last_slab()[:, :ws]
        frame32 = pipe.last_slab_f32()[:, :ws]
    elif pipe.rotate_root:                              # the last batch was assembled on its own root: rank 0 checks it
        last_root = (pipe.batches - 1) % world
        buf = frame.clone() if rank == last_root else torch.empty((3, w, h), dtype=torch.uint8, device=dev)
        dist.broadcast(buf, src=last_root)
        frame = buf
        if a.gather == "f32":
            buf32 = frame32.clone() if rank == last_root else torch.empty((3, w, h), dtype=torch.float32, device=dev)
            dist.broadcast(buf32, src=last_root)
            frame32 = buf32

    mine = torch.tensor([dt, kernel_ms, (t_submitted - t0) / a.steps * 1e3, float(x0), float(x1)], dtype=torch.float64, device=dev)
    per_rank = None
    if world > 1:                                       # every rank's own numbers travel to rank 0's line
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = [{"rank": q, "columns": [int(v[3]), int(v[4])], "wall_ms_per_step": round(float(v[0]) / a.steps * 1e3, 5),
                     "launch_ms": round(float(v[1]), 5), "host_submit_ms_per_step": round(float(v[2]), 5)} for q, v in enumerate(allv)]
        dt, kernel_ms_max = max(float(v[0]) for v in allv), max(float(v[1]) for v in allv)
    else:
        kernel_ms_max = kernel_ms

    if rank == 0:
        rays = wl["rays"]
        if rays is None:   # ray counts of non-headline configs come from the oracle's counters
            from oracle import oracle as orc
            c = orc.render(w, h, cam.position, cam.rotation, wl["spheres"], wl["lights"], wl["planes"], wl["amb"], wl["lamb"],
                           wl["refl"], wl["depth"], wl["aa"], raygen=cam.raygen(), want=(), spp=wl["spp"], seed=wl["seed"])["counters"]
            rays = dict(closest=c["closest"], shadow=c["shadow"])
        rays_per_frame = rays["closest"] + rays["shadow"]
        ms_per_step = dt / a.steps * 1e3
        # algorithmic HBM bytes of one launch of this rank's kernel: float32 RGB planes (12 B/px) + uint8
        # frame (3 B/px) stored once, scene + camera read once (DESIGN.md "Measurement")
        S, L, P = wl["spheres"].shape[1], wl["lights"].shape[1], wl["planes"].shape[1]
        alg_bytes = ws * h * 15 + 4 * (7 * S + 3 * L + 9 * P) + 96      # per frame
        fpl = frames_per_launch_avg                      # frames one launch of the timed region rendered (mean)
        alg_bytes_launch = alg_bytes * fpl
        kernel_eff = kernel_ms_max / NS / fpl            # time per frame: NS launches of fpl frames each are in flight at a time
        achieved = alg_bytes / (kernel_eff * 1e-3) / 1e9
        so_sha = hashlib.sha256(open(_lib.SO_PATH, "rb").read()).hexdigest()
        headline1 = world == 1 and name == workloads.HEADLINE
        suffix = "" if name == workloads.HEADLINE else f"_{name}"
        tr = stamped(os.path.join(REPO, "profiles", f"traffic_{ROUND}{suffix}.json"), so_sha) if world == 1 else None
        traffic = tr.get("hbm_bytes_per_launch") if tr else None
        frame_host = frame.cpu().numpy()
        check = check32 = None
        gpath = os.path.join(REPO, "tests", "golden", "frame_c2_1080p.npz")
        if name == workloads.HEADLINE and os.path.exists(gpath):
            gold = np.load(gpath)
            check = hashlib.sha256(frame_host.tobytes()).hexdigest() == str(gold["sha256_u8"])
            # the float32 pre-clip planes (north_star's tolerance is stated on them): this rank's own at N = 1, the ASSEMBLED
            # frame with --gather f32
            if frame32 is not None and tuple(frame32.shape) == (3, w, h):
                check32 = hashlib.sha256(np.ascontiguousarray(frame32.cpu().numpy()).tobytes()).hexdigest() == str(gold["sha256_rgb32"])

        # -- what the kernel really traced (counting instantiation, one frame, outside the timed region)
        traced = None
        if world == 1:
            r.reset_stats()
            pc = r.params(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], wl["aa"], spp=wl["spp"], seed=wl["seed"],
                          flags=_lib.RT_FLAG_COUNT_RAYS)
            r.render_device(pc, x0, x1, pipe.u8v[0][0].data_ptr(), pipe.f32v[0][0].data_ptr(), PS, pipe.stream_handle(0))
            torch.cuda.synchronize()
            st = r.stats()
            traced = {"closest_queries": st["closest_queries"], "shadow_traced": st["shadow_traced"],
                      "shadow_skipped_answer_unused": st["shadow_skipped"], "hits": st["hits"],
                      "total_traced": st["closest_queries"] + st["shadow_traced"]}

        # -- the same frames on ONE stream, strictly one after another: per-launch duration = frame period, so the
        # roofline fraction follows from the plain formula (bytes / duration / peak)
        serial = None
        if world == 1 and not a.no_serial:
            ns = max(a.steps, 200)
            sh = pipe.stream_handle(0)
            u8p, f32p = pipe.u8v[0][0].data_ptr(), pipe.f32v[0][0].data_ptr()
            for _ in range(20):
                r.render_device(params, x0, x1, u8p, f32p, PS, sh)
            torch.cuda.synchronize()
            se = [torch.cuda.Event(enable_timing=True) for _ in range(ns + 1)]
            se[0].record(pipe.streams[0])
            for i in range(ns):
                r.render_device(params, x0, x1, u8p, f32p, PS, sh)
                se[i + 1].record(pipe.streams[0])
            torch.cuda.synchronize()
            durs = [se[i].elapsed_time(se[i + 1]) for i in range(ns)]
            mean = se[0].elapsed_time(se[ns]) / ns
            serial = {"launches": ns, "kernel_ms": round(mean, 5), "kernel_ms_median": round(statistics.median(durs), 5),
                      "kernel_ms_min": round(min(durs), 5), "achieved": round(alg_bytes / (mean * 1e-3) / 1e9, 3),
                      "frac": round(alg_bytes / (mean * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                      "note": "one stream, ONE frame per launch, launches strictly back to back (round 2's serial mode): achieved = "
                              "algorithmic_bytes_per_frame / kernel_ms, the plain formula"}

        # -- the same frames with a camera that moves with EVERY frame (README.md:23 "real-time display"; scene/camera.py:8-16):
        # one rt_render_sequence call per chunk of frames carries their cameras; one launch per frame, round-robin on the
        # streams; the dispatch order measured under an earlier camera is kept and refreshed every MI355RT_REMEASURE frames,
        # the cull tables are rebuilt per camera position on the launching stream.  Same timing discipline as the main
        # region (fence, K steps, fence); never `value`.
        dynamic = None
        if world == 1 and not a.no_dynamic:
            DNS = 3                                     # one launch per frame: three streams let consecutive frames overlap
            dstreams = pipe.streams if NS >= DNS else [torch.cuda.Stream(device=dev) for _ in range(DNS)]
            chunk = 8 * len(dstreams)                   # frames per call; frame i of a chunk always lands on stream i % DNS
            cams = workloads.camera_path(chunk * 16)    # the path repeats; consecutive frames always differ
            dyn8 = torch.empty((chunk, 3, ws, h), dtype=torch.uint8, device=dev)
            dyn32 = torch.empty((chunk, 3, ws, h), dtype=torch.float32, device=dev)
            handles = tuple(st_.cuda_stream for st_ in dstreams)

            def dyn(nframes, start=0):
                done_ = 0
                while done_ < nframes:
                    nf = min(chunk, nframes - done_)
                    c0 = ((start + done_) // chunk % 16) * chunk
                    r.render_sequence(params, x0, x1, nf, dyn8.data_ptr(), dyn32.data_ptr(), ws * h, 3 * ws * h,
                                      cams[c0:c0 + nf], handles, 0)
                    done_ += nf
            dyn(max(a.warmup, 4 * chunk))
            dsteps = max(a.steps, 10 * chunk)           # one launch per frame: a short region would time the pipeline's fill and drain
            fence()
            td = time.perf_counter()
            dyn(dsteps, start=chunk)
            td_sub = time.perf_counter()
            fence()
            dyn_ms = (time.perf_counter() - td) / dsteps * 1e3
            st = r.stats()
            dynamic = {"ms_per_step": round(dyn_ms, 5), "ratio_to_static": round(dyn_ms / ms_per_step, 4),
                       "host_submit_ms_per_step": round((td_sub - td) / dsteps * 1e3, 5), "streams": len(dstreams), "steps": dsteps,
                       "camera": "python_ray_tracer_amd.workloads.camera_path: position, pitch, yaw and roll change with every frame",
                       "note": "one launch per frame (every frame has its own camera and cull tables); frames bit-equal to the oracle: "
                               "tests/test_gpu_parity.py::test_moving_camera_sequence"}
            r.set_camera(cam.position, cam.rotation)    # back to the static camera for what follows
            if dstreams is not pipe.streams:
                for st_ in dstreams:
                    r.stream_forget(st_.cuda_stream)
            del dyn8, dyn32

        # -- the host-buffer entry point (what an unchanged main.py sees: launch + copy_to_host); never `value`
        host_path = None
        if world == 1 and not a.no_host_path:
            host_path = host_path_ms(r, wl, np)

        out = {
            "metric": "Mrays/sec + frame time (ms) at 1920x1080, 8 spheres, depth=3",
            "value": round(rays_per_frame / (dt / a.steps) / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 5),
            "rehearsal": "all ranks on one GPU, collectives on gloo through host copies: the timings are NOT measurements" if a.rehearse_gloo else None,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": name, "width": w, "height": h, "spheres": S, "planes": P, "lights": L,
                       "depth": wl["depth"], "aa": bool(wl["aa"]), "rays_per_frame": rays_per_frame,
                       "rays_per_frame_is": "the reference algorithm's query count for this frame (closest-hit + shadow queries, "
                                            "oracle counters); rays_traced_per_frame is what the kernel traces",
                       "rays_traced_per_frame": traced["total_traced"] if traced else None,
                       "primary_rays_per_frame": w * h, "outputs": "uint8 (3,w,h) frame + float32 (3,w,h) pre-clip RGB",
                       "streams": NS,
                       "parallelism": f"column slabs x{world}, frames queued round-robin on {NS} stream(s)" + (f", one RCCL gather of the uint8 slabs of {F} frames per {F} steps to " + ("rank (batch mod N) — every frame is assembled on one rank, the ranks take turns" if pipe.rotate_root else "rank 0") + ", overlapped with the next steps' renders" if use_gather else "")},
            "frame_ms": round(ms_per_step, 5),
            "frame_ms_median": round(statistics.median(periods), 5) if periods else None,
            "frame_ms_min": round(min(periods), 5) if periods else None,
            "frame_latency_ms": round(kernel_ms_max, 5),
            "launch_ms_median": round(statistics.median(per_launch), 5) if per_launch else None,
            "launch_ms_min": round(min(per_launch), 5) if per_launch else None,
            "preheat_ms": round(preheat_ms, 1), "preheat_frames": preheat_frames,
            "host_submit_ms_per_step": round((t_submitted - t0) / a.steps * 1e3, 5),
            "gpu_span_ms": round(gpu_span_ms, 4), "timed_region_ms": round(dt * 1e3, 4),   # device clock (first launch's start to the last one's end) / host clock between the fences
            "primary_mrays_per_s": round(w * h / (dt / a.steps) / 1e6, 2),
            "traced_mrays_per_s": round(traced["total_traced"] / (dt / a.steps) / 1e6, 2) if traced else None,
            "rays_traced": traced,
            "value_is": "reference-algorithm scene queries (closest-hit + shadow, SURVEY.md section 8d 'total') answered per second; "
                        "traced_mrays_per_s counts only the queries the kernel traces (it skips shadow queries whose answer the "
                        "reference discards, trace.py:101)",
            "frames_per_launch": round(fpl, 3), "launches": len(per_launch) if per_launch else None,
            "per_rank": per_rank,
            "frame_matches_reference_sha256": check,
            "float32_frame_matches_reference_sha256": check32,
            "slab_balance": balance,
            "dynamic": dynamic,
            "host_path_ms": host_path,
            "library_sha256": so_sha,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "kernel": "rt::render_kernel", "kernel_ms": round(kernel_ms_max, 5), "launches_in_flight": NS,
                         "frames_per_launch": round(fpl, 3), "algorithmic_bytes_per_frame": alg_bytes,
                         "algorithmic_bytes": int(alg_bytes_launch), "ms_per_frame_effective": round(kernel_eff, 5),
                         "per_launch": {"achieved": round(alg_bytes_launch / (kernel_ms_max * 1e-3) / 1e9, 3),
                                        "frac": round(alg_bytes_launch / (kernel_ms_max * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                        "note": "plain formula on the launches of the timed region as they ran (launches_in_flight of them "
                                                "side by side): algorithmic_bytes / kernel_ms"},
                         "wall": {"achieved": round(alg_bytes / (ms_per_step * 1e-3) / 1e9, 3),
                                  "frac": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                  "note": "algorithmic_bytes_per_frame / ms_per_step: what a clock around the whole timed region sees"},
                         "serial": serial,
                         "note": "kernel_ms = mean duration of one launch (HIP event behind every launch, on its stream; what "
                                 "rocprofv3's kernel trace shows); a launch renders frames_per_launch frames (algorithmic_bytes = "
                                 "frames_per_launch x algorithmic_bytes_per_frame) and launches_in_flight launches run side by side, so "
                                 "achieved = launches_in_flight x algorithmic_bytes / kernel_ms (= bytes per frame period); with "
                                 "--streams 1 that IS the plain formula, and `per_launch`, `wall` and `serial` follow from it at any "
                                 "setting.  float64 VALU-bound by construction (about 15 B and ~2 kflop per pixel): the HBM fraction is "
                                 "reported because BASELINE.json's north_star asks for it, not because HBM limits this kernel"},
        }
        # The bound that actually limits this kernel: VALU instruction issue.  Per-launch counts from the rocprofv3 op-mix
        # pass of THIS library build (profiles/valu_<round>[_<workload>].json carries the build's SHA-256; another build ->
        # null), priced with the cycles per wave-instruction tools/valu_prices.hip measured (profiles/r03_valu_prices.json;
        # MI355X_MICROARCH.md, "Per-instruction cycle constants": wave64 v_fma_f32 2 cycles with co-resident waves, 4 alone).
        v = stamped(os.path.join(REPO, "profiles", f"valu_{ROUND}{suffix}.json"), so_sha) if world == 1 else None
        ppath = os.path.join(REPO, "profiles", PRICES)
        if v and os.path.exists(ppath):
            info = r.kernel_info()
            # launch bounds of the profiled instantiation: the LDS-parked variants (second template argument) are compiled for 7
            # waves per SIMD, the lane-owned kernels (MODE 2) for 4
            kn = str(v.get("kernel") or "")
            waves = 7 if (", true, " in kn.split("<")[-1][:14] and not kn.rstrip(">(rt::KParams)").endswith("2")) else 4
            lo, est, table, col = issue_bound(v, json.load(open(ppath)), waves)
            util = v.get("lane_utilisation") or 1.0
            flop = v["fp64_flop_wave_level_x64"] * util                 # wave-level count x 64 lanes x live-lane fraction
            out["valu"] = {"valu_wave_instructions_per_launch": v["valu_wave_instructions_per_launch"],
                           "salu_wave_instructions_per_launch": v.get("salu_wave_instructions_per_launch"),
                           "lane_utilisation": v.get("lane_utilisation"),
                           "fp64_flop_per_launch": int(flop),
                           "achieved_fp64_tflops": round(flop / (kernel_eff * 1e-3) / 1e12, 3),
                           "peak_fp64_vector_tflops": 78.6, "frac_of_fp64_peak": round(flop / (kernel_eff * 1e-3) / 78.6e12, 4),
                           "issue_bound_ms": round(lo, 5), "issue_frac": round(lo / kernel_eff, 4),
                           "issue_estimate_ms": round(est, 5), "issue_estimate_frac": round(est / kernel_eff, 4),
                           "prices": {"file": f"profiles/{PRICES}", "column": col, "clock_ghz": CLOCK_HZ / 1e9, "simds": SIMDS, "classes": table},
                           "kernel_vgprs": info.get("vgprs"),
                           "note": "issue_bound_ms = sum over classes of (wave-instructions x measured cycles per wave-instruction at this "
                                   "occupancy) / (1024 SIMDs x 2.4 GHz), unclassified instructions at the cheapest measured price: the launch "
                                   "cannot be shorter at this instruction count; issue_estimate_ms prices them as compares/selects (4.1 "
                                   "cycles).  issue_frac = issue_bound_ms / time per launch (<= 1 by construction of a bound).  "
                                   "The scalar unit (one per CU, ~1 instruction per cycle) is the second bound: "
                                   "salu_wave_instructions_per_launch / (256 x 2.4 GHz)."}
            if v.get("salu_wave_instructions_per_launch"):
                out["valu"]["salu_bound_ms"] = round(v["salu_wave_instructions_per_launch"] * 1.09 / (256 * CLOCK_HZ) * 1e3, 5)
            # the "reference-equivalent" flop figure of the frame (the work the reference's formulation
            # would do for the same queries: 18 S + 14 P per scene query, 51 + 24 L per shaded hit)
            if traced:
                out["valu"]["reference_equivalent_flop_per_frame"] = int(rays_per_frame * (18 * S + 14 * P) + traced["hits"] * (51 + 24 * L))
        else:
            out["valu"] = None
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, rays_per_frame)
            # SURVEY.md section 8(d) / BASELINE.md section 2: the reference's own Python (numba.cuda kernels run as plain Python
            # under an identity-decorator stand-in, the arithmetic NUMBA_ENABLE_CUDASIM=1 runs) cannot travel to the GPU box; its
            # speed was measured once in the build container and is quoted as a constant with its provenance
            out["cpu_baseline"]["reference_python"] = {
                "value": 0.045, "unit": "Mrays/s", "cores": 1, "frame": "config 1: 128x128, 3 spheres + plane, depth 1",
                "frame_s": 2.34, "rays_per_frame": 105404,
                "provenance": "BASELINE.md section 2 / SURVEY.md section 8(d) [probe]: 2.34 s per config-1 frame on one core of the build "
                              "container's 8-core Xeon @ 2.1 GHz (143 us per pixel); a constant, not measured in this run"}
        print(json.dumps(out), flush=True)
    pipe.close(r)
    r.close()
    if world > 1 or a.force_gather:
        dist.barrier()
        dist.destroy_process_group()


def balance_slabs(r, params, w, h, world, rank, dev, dist, torch, ns, rounds, frames=400):
    """Every rank times its own slab (frames round-robin on `ns` streams, as in the run), the times are all-gathered
    and python_ray_tracer_amd.distributed.SlabBalancer moves the boundaries; the last round only measures."""
    from python_ray_tracer_amd.distributed import SlabBalancer
    bal = SlabBalancer(w, world)
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    u8 = [torch.empty(3 * w * h, dtype=torch.uint8, device=dev) for _ in range(ns)]
    f32 = [torch.empty(3 * w * h, dtype=torch.float32, device=dev) for _ in range(ns)]
    history = []
    for i in range(600):                                  # clocks up before anything is compared (about 10-40 ms)
        a_, b_ = bal.bounds[rank]
        r.render_device(params, a_, b_, u8[i % ns].data_ptr(), f32[i % ns].data_ptr(), (b_ - a_) * h, streams[i % ns].cuda_stream)
    torch.cuda.synchronize()
    for it in range(rounds + 1):
        a_, b_ = bal.bounds[rank]

        def burst(n):
            for i in range(n):
                r.render_device(params, a_, b_, u8[i % ns].data_ptr(), f32[i % ns].data_ptr(), (b_ - a_) * h, streams[i % ns].cuda_stream)
        burst(2); torch.cuda.synchronize()                # measure the tile costs, build the dispatch order
        burst(2 * ns); torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        burst(frames)
        torch.cuda.synchronize()
        mine = torch.tensor([(time.perf_counter() - t0) / frames * 1e3], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        times = [float(t[0]) for t in allt]
        history.append({"bounds": [list(b) for b in bal.bounds], "slab_ms": [round(t, 5) for t in times],
                        "max_over_mean": round(max(times) / (sum(times) / world), 4)})
        if it < rounds:
            bal.update(times)
    for s_ in streams:
        r.stream_forget(s_.cuda_stream)
    best = min(range(len(history)), key=lambda i: history[i]["max_over_mean"])
    bounds = [tuple(b) for b in history[best]["bounds"]]
    return bounds, {"equal_width": history[0], "chosen": history[best], "rounds": rounds}


def host_path_ms(r, wl, np, frames=40):
    """rt_render into host memory (launch + device-to-host copy + sync per frame), uint8 frame only and with the
    float32 buffer as well; pageable numpy arrays and pinned arrays from the library (Renderer.host_array)."""
    out = {}
    for label, want32 in (("u8", False), ("u8_f32", True)):
        for mem in ("pageable", "pinned"):
            bufs = r.host_arrays(want32, pinned=(mem == "pinned"))
            for _ in range(3):
                r.render_into(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], wl["aa"], *bufs, spp=wl["spp"], seed=wl["seed"])
            t0 = time.perf_counter()
            for _ in range(frames):
                r.render_into(wl["amb"], wl["lamb"], wl["refl"], wl["depth"], wl["aa"], *bufs, spp=wl["spp"], seed=wl["seed"])
            out[f"{label}_{mem}"] = round((time.perf_counter() - t0) / frames * 1e3, 4)
            r.release_host_arrays(bufs)
    # a sequence of frames through rt_render_begin / rt_render_end: three slots, each with its own pinned arrays; the
    # copy of one frame overlaps the rendering of the next (every frame still arrives complete in host memory)
    for label, want32 in (("u8", False), ("u8_f32", True)):
        slots = 3
        bufs = [r.host_arrays(want32, pinned=True) for _ in range(slots)]
        def seq(n):
            for i in range(n):
                if i >= slots:
                    r.render_end(i % slots)
                r.render_begin(i % slots, wl["amb"], wl["lamb"], wl["refl"], wl["depth"], wl["aa"], *bufs[i % slots], spp=wl["spp"], seed=wl["seed"])
            for sl in range(slots):
                r.render_end(sl)
        seq(6)
        t0 = time.perf_counter()
        seq(3 * frames)
        out[f"{label}_pinned_sequence"] = round((time.perf_counter() - t0) / (3 * frames) * 1e3, 4)
        for b in bufs:
            r.release_host_arrays(b)
    out["note"] = ("ms per frame, PCIe-inclusive; never `value`.  *_pageable / *_pinned: the synchronous rt_render (kernel + D2H + sync "
                   "per call); *_pinned_sequence: frames in flight on 3 slots (rt_render_begin / rt_render_end), every frame complete in host memory")
    return out


if __name__ == "__main__":
    main()
