"""RCCL-side checks that fit a one-GPU box: the FrameGatherer call pattern (async gather into views of a
staging buffer, strided assembly copy, double-buffering) on the `nccl` backend with world_size 1, fed by
the HIP kernel on the stream torch considers current.  Multi-rank assembly is covered on CPU (gloo)."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import os, sys, hashlib
sys.path.insert(0, %(repo)r); sys.path.insert(0, os.path.join(%(repo)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import python_ray_tracer_amd as pkg
from python_ray_tracer_amd import workloads
from python_ray_tracer_amd.distributed import FrameGatherer, slab_bounds
from conftest import load_frame
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
wl = workloads.build("c1_128x128_s3_d1"); cam = wl["camera"]; w, h = wl["w"], wl["h"]
r = pkg.Renderer(0)
r.set_scene(wl["spheres"], wl["lights"], wl["planes"]); r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *cam.raygen())
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
g = FrameGatherer(w, h, torch.uint8, dev, dist, dst=0, slots=2)
slabs = [torch.zeros((3, w, h), dtype=torch.uint8, device=dev) for _ in range(2)]
frames = {}
busy = [None, None]
for i in range(4):
    b = i %% 2
    if busy[b] is not None:
        frames[busy[b]] = g.finish(b).cpu().numpy().copy()
    p = r.params(wl["amb"], wl["lamb"], wl["refl"], i %% 2, False)      # depth alternates 0/1
    r.render_device(p, 0, w, slabs[b].data_ptr(), None, w * h, s.cuda_stream)
    g.submit(slabs[b], b); busy[b] = i
for b in range(2):
    frames[busy[b]] = g.finish(b).cpu().numpy().copy()
torch.cuda.synchronize()
gold = load_frame("c1_128")["frame_u8"]
assert np.array_equal(frames[1], gold) and np.array_equal(frames[3], gold), "depth-1 frames differ from the golden frame"
assert np.array_equal(frames[0], frames[2]) and not np.array_equal(frames[0], frames[1])
r.close(); dist.destroy_process_group()
print("NCCL_GATHER_OK")
"""


def test_frame_gatherer_on_nccl_world1():
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"repo": REPO}], capture_output=True, text=True, timeout=600)
    assert "NCCL_GATHER_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_bench_runs_under_torchrun_world1():
    """bench.py through the launcher the driver uses for N>1 (here N=1): one JSON line, frame hash ok."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29544", os.path.join(REPO, "bench.py"),
                          "--gpus", "1", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["frame_matches_reference_sha256"] is True and j["roofline"]["frac"] > 0


@pytest.mark.parametrize("extra", [["--frames-per-gather", "4", "--steps", "22"], ["--frames-per-gather", "1", "--streams", "2", "--steps", "9"]])
def test_bench_exchange_structure_on_one_gpu(extra):
    """The N > 1 choreography of bench.py (frames round-robin on render streams, slabs of F frames per RCCL gather
    on a separate stream, two exchanges in flight, last batch partly filled) on a world-size-1 RCCL group."""
    import json
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--force-gather", "--warmup", "5", "--no-cpu-baseline"] + extra,
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29546"))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["frame_matches_reference_sha256"] is True and "gather" in j["config"]["parallelism"]


def test_bench_force_gather_float32_assembly():
    """--gather f32 on a world-size-1 RCCL group: the float32 pre-clip planes travel through their own gather and the
    assembled frame's hash is still the reference's (the uint8 one); bench.py's line carries the batched-launch fields."""
    import json
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--force-gather", "--gather", "f32", "--warmup", "5", "--steps", "20",
                          "--no-cpu-baseline", "--no-host-path", "--no-serial", "--no-dynamic"],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547"))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["frame_matches_reference_sha256"] is True and j["float32_frame_matches_reference_sha256"] is True
    assert j["frames_per_launch"] > 1           # (host_submit_ms_per_step includes waiting for a slot's previous exchange here)


def test_bench_default_line_fields():
    """The default single-GPU line under the driver's flags: batched launches (host submit cost far below a frame), the
    moving-camera block, the priced issue bound when the profile constants match this build."""
    import json
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-host-path"],
                         capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["frame_matches_reference_sha256"] is True and j["config"]["streams"] == 1 and j["frames_per_launch"] >= 4
    assert j["host_submit_ms_per_step"] <= 0.02
    d = j["dynamic"]
    assert d and d["ms_per_step"] > 0 and d["steps"] >= 20
    assert j["roofline"]["launches_in_flight"] == 1 and abs(j["roofline"]["frac"] - j["roofline"]["per_launch"]["frac"]) < 1e-9
    if j["valu"]:
        assert j["valu"]["issue_frac"] <= 1.0 and j["valu"]["issue_bound_ms"] < j["valu"]["issue_estimate_ms"]


def test_bench_spawns_its_rank_for_config5_spp4():
    """`python bench.py --gpus 1 --workload c5_..._spp4` through torch.distributed.run (the launcher the driver uses for
    N > 1), a few steps of the largest BASELINE configuration: one JSON line with the workload's ray counts."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29548", os.path.join(REPO, "bench.py"),
                          "--gpus", "1", "--workload", "c5_7680x4320_s256_d8_spp4", "--steps", "4", "--warmup", "1", "--preheat-ms", "0",
                          "--no-cpu-baseline", "--no-host-path", "--no-serial", "--no-dynamic"],
                         capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["config"]["workload"] == "c5_7680x4320_s256_d8_spp4" and j["config"]["rays_per_frame"] == 427049977 + 896048151
    assert j["rays_traced"]["closest_queries"] == 427049977 and 10.0 < j["ms_per_step"] < 200.0


@pytest.mark.parametrize("n,extra", [(2, ["--gather", "f32"]), (3, ["--gather-root", "fixed", "--frames-per-gather", "4"])])
def test_bench_multi_rank_rehearsal_on_one_gpu(n, extra):
    """bench.py's N > 1 control flow — slab balancing from measured times, batches of frames gathered to a rotating (or fixed)
    root, float32 assembly, per-rank statistics — with N processes sharing this box's one GPU (--rehearse-gloo: RCCL refuses two
    ranks on a device, so the collectives run on gloo through host copies).  The assembled frame of the last batch must be the
    reference's; every rank must appear in the line.  Timings of such a run are not measurements and are not checked."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                          "--master-addr", "127.0.0.1", "--master-port", str(29560 + n), os.path.join(REPO, "bench.py"),
                          "--gpus", str(n), "--rehearse-gloo", "--steps", "24", "--warmup", "4", "--preheat-ms", "20", "--balance-rounds", "2",
                          "--no-cpu-baseline", "--no-serial", "--no-dynamic"] + extra,
                         capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["rehearsal"] and j["frame_matches_reference_sha256"] is True
    if "f32" in extra:
        assert j["float32_frame_matches_reference_sha256"] is True
    assert len(j["per_rank"]) == n and [q["rank"] for q in j["per_rank"]] == list(range(n))
    cols = [q["columns"] for q in j["per_rank"]]
    assert cols[0][0] == 0 and cols[-1][1] == j["config"]["width"] and all(cols[i][1] == cols[i + 1][0] for i in range(n - 1))
    assert j["slab_balance"]["rounds"] == 2 and len(j["slab_balance"]["chosen"]["slab_ms"]) == n
