"""The N>1 PRODUCT path on real kernels: two ranks (both on cuda:0 — the GPU box has one device; Gloo rendezvous)
shard a video list through CapFiltEngine + VisualTokenizer + the two write_outputs() and must produce the same three
JSON files, byte for byte, as one process handling all videos (run_video_CapFilt.py:237-291,
run_visual_tokenization.py:427-463).  Also smokes `bench.py --gpus 2` in its one-device mode.
Where the node has at least two GPUs the same comparison runs with ONE RANK PER DEVICE over RCCL (backend "nccl":
gather_json's device-buffer send / recv, barrier(device_ids=...), utils.py:258-281) — skipped on a one-GPU box."""
import json
import os
import socket
import subprocess
import sys

import pytest

from common import ROOT

pytestmark = pytest.mark.gpu

WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch
from common import synthetic_frames
from vidil_amd import dist as vdist, capfilt, visual_tokenization as vt
from vidil_amd.blip import BLIP_Decoder
from vidil_amd.blip_itm import BLIP_ITM
from vidil_amd.clip import CLIPModel
from vidil_amd.tokenizer import SyntheticBertTokenizer

rank, world, local = vdist.init_distributed_mode(backend={backend!r})
devi = local if {backend!r} == "nccl" else 0
torch.cuda.set_device(devi)
dev = torch.device("cuda", devi)
torch.manual_seed(0)
tok = SyntheticBertTokenizer()
cap = BLIP_Decoder(image_size=224, vit="base", tokenizer=tok).eval()
itm = BLIP_ITM(image_size=224, vit="base", tokenizer=tok).eval()
clip = CLIPModel().eval()
cfg = dict(caption=True, filter=True, filter_generated_only=True, keep_original_caption=False, threshold=0.4,
           filter_mode="max_filter", generation_mode="beam", do_sentence_tokenization=False, image_size=224, vit="base",
           topk_visualize=5)
NV, F = 5, 4
g = torch.Generator().manual_seed(3)
sizes = dict(objects=500, attributes=300, scenes=65, verbs=96)
emb = {{k: torch.nn.functional.normalize(torch.randn(n, 512, generator=g), dim=-1) for k, n in sizes.items()}}
texts = {{k: [f"{{k}}{{i}}" for i in range(n)] for k, n in sizes.items()}}
eng = capfilt.CapFiltEngine(cfg, dev, captioner=cap, filterer=itm)
vtk = vt.VisualTokenizer(cfg, clip, texts, emb, dev)
videos = [f"video{{i}}" for i in range(NV)]
s, e = vdist.shard_bounds(NV)
items = [dict(video_id=v, text=[]) for v in videos[s:e]]
toks = {{}}
if e > s:
    u8 = torch.from_numpy(synthetic_frames(e - s, F, first_video=s)).to(dev)
    eng.process(items, u8)
    toks = vtk.process([it["video_id"] for it in items], u8, [it["unfiltered_text"] for it in items])
f, u = capfilt.collect_outputs(items)
capfilt.write_outputs({out!r}, f, u)
vt.write_outputs({out!r}, toks)
seen = vdist.ranks_seen()
# (gloo: one identity per process; nccl: one per DEVICE — equal to the world size exactly when every rank has its own GPU)
assert seen == (world if {backend!r} == "gloo" else len({{str(getattr(torch.cuda.get_device_properties(i), 'uuid', i)) for i in range(world)}})), seen
vdist.barrier()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_RENDEZVOUS_ERRORS = ("Address already in use", "Connection refused", "Connection reset", "connect() timed out",
                      "failed to connect", "EADDRINUSE")


def _run(world, out, backend="gloo"):
    script = WORKER.format(root=ROOT, out=out, backend=backend)
    for attempt in range(3):         # a lost race for the probed port is retried on a fresh one; anything else fails at once
        port = _free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r if backend == "nccl" else 0), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [(p, p.communicate(timeout=900)[0].decode()) for p in procs]
        bad = [o[-4000:] for p, o in outs if p.returncode != 0]
        if not bad:
            return
        if attempt == 2 or not any(e in o for o in bad for e in _RENDEZVOUS_ERRORS):
            raise AssertionError("\n".join(bad))


@pytest.fixture(scope="module")
def single_process_out(tmp_path_factory):
    """The un-distributed run's three files, once per module (every comparison below is against them)."""
    out = str(tmp_path_factory.mktemp("single") / "w1")
    _run(1, out)
    return out


def test_two_ranks_on_one_gpu_through_the_engines_equal_single_process(tmp_path, single_process_out):
    out1, out2 = single_process_out, str(tmp_path / "w2")
    _run(2, out2)
    for name in ("video_text_CapFilt.json", "video_text_Cap.json", "visual_tokens.json"):
        a = open(os.path.join(out1, name)).read()
        b = open(os.path.join(out2, name)).read()
        assert a == b, name
    caps = json.load(open(os.path.join(out2, "video_text_Cap.json")))
    assert list(caps.keys()) == [f"video{i}" for i in range(5)]
    toks = json.load(open(os.path.join(out2, "visual_tokens.json")))
    assert list(toks.keys()) == [f"video{i}" for i in range(5)] and len(toks["video4"]["frame_tokens"]) == 4


def test_rccl_backend_with_a_single_rank_runs_the_device_side_collectives(tmp_path, single_process_out):
    """What a one-GPU box CAN execute of the RCCL branch: backend "nccl" with world size 1 — process-group creation on the
    device, `barrier(device_ids=...)`, BOTH all_gathers of gather_json (sizes, then the padded payload bytes: since round 4
    the gather has no peer-to-peer half, so the code an 8-GPU job takes is exactly the code that runs here) and the
    max-reduce of `vidil_amd.dist` on DEVICE buffers.  Outputs equal the un-distributed run's."""
    out0, out1 = single_process_out, str(tmp_path / "nccl1")
    _run(1, out1, backend="nccl")
    for name in ("video_text_CapFilt.json", "video_text_Cap.json", "visual_tokens.json"):
        assert open(os.path.join(out0, name)).read() == open(os.path.join(out1, name)).read(), name
    probe = r"""
import os, sys, torch
sys.path.insert(0, %r)
from vidil_amd import dist as vdist
rank, world, local = vdist.init_distributed_mode(backend="nccl")
assert (rank, world) == (0, 1) and torch.distributed.get_backend() == "nccl"
assert vdist._comm_device().type == "cuda"
assert vdist.gather_json({"a": [1, 2, 3]}) == [{"a": [1, 2, 3]}]
assert vdist.max_over_ranks(1.25) == 1.25
assert vdist.ranks_seen() == 1          # (round 6: the device identity — GPU UUID — all_gather'ed as a DEVICE buffer over RCCL)
os.environ["VIDIL_GATHER"] = "allgather"
assert vdist.gather_json({"b": 2}) == [{"b": 2}]          # (the other collective form, chosen before it is issued)
vdist.barrier()
print("nccl-single-rank-ok")
""" % ROOT
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", probe], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "nccl-single-rank-ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_one_rank_per_gpu_over_rccl_equals_single_process(tmp_path, single_process_out):
    """The RCCL branch of vidil_amd.dist (backend "nccl": sizes, then the padded JSON bytes all_gather'ed as DEVICE
    buffers over xGMI, barrier pinned to the rank's device): 2 ranks (4 when the node has them), one per GPU, through both
    engines and both writers; the merged files equal the single-process ones byte for byte."""
    import torch

    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"needs >= 2 GPUs for one rank per device over RCCL (this node has {n})")
    out1 = single_process_out
    for world in sorted({2, min(n, 4)}):
        outn = str(tmp_path / f"w{world}_nccl")
        _run(world, outn, backend="nccl")
        for name in ("video_text_CapFilt.json", "video_text_Cap.json", "visual_tokens.json"):
            assert open(os.path.join(out1, name)).read() == open(os.path.join(outn, name)).read(), (world, name)


def test_bench_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` with N > the node's device count and no one-device smoke flag must refuse (a silent pile-up of
    ranks on cuda:0 would report a scaling number that is not one)."""
    import torch

    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k != "VIDIL_BENCH_SMOKE_ONE_DEVICE"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "one rank per GPU" in (r.stderr + r.stdout)


@pytest.mark.slow      # (test_bench_launches_itself_for_gpus_1_and_gpus_2 runs the same two ranks through the launcher)
def test_bench_gpus_2_one_device_smoke(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one JSON line from rank 0), with both
    ranks on the box's single GPU (VIDIL_BENCH_SMOKE_ONE_DEVICE=1 -> Gloo).  The log is kept under gpurun_out/."""
    env = dict(os.environ, VIDIL_BENCH_SMOKE_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--videos-per-step", "16"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert "roofline" not in rec and "cpu_baseline" not in rec        # rank 0 at N=1 only
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_gpus2_one_device_smoke.log"), "w") as f:
        f.write("$ VIDIL_BENCH_SMOKE_ONE_DEVICE=1 " + " ".join(cmd[1:]) + "\n" + r.stdout + "\n---- stderr ----\n" + r.stderr[-3000:])


def _json_lines(text):
    return [l for l in text.splitlines() if l.startswith("{")]


def test_bench_launches_itself_for_gpus_1_and_gpus_2():
    """`python bench.py --gpus N` exactly as the driver types it — no RANK in the environment, no external
    launcher.  N = 1 runs in-process; N = 2 makes bench.py re-run itself under torch.distributed.run (one-device smoke flag
    on this 1-GPU box: both ranks on cuda:0 over Gloo; on a multi-GPU node the same command is one rank per GPU over RCCL).
    One JSON line from rank 0, n_gpus = N, exit code 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    common = ["--steps", "1", "--warmup", "0", "--videos-per-step", "16", "--no-cpu-baseline", "--no-secondary"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1
    import torch
    if torch.cuda.device_count() < 2:
        env["VIDIL_BENCH_SMOKE_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_gpus2_self_launch.log"), "w") as f:
        f.write("$ python bench.py --gpus 2 " + " ".join(common) + "\n" + r.stdout + "\n---- stderr ----\n" + r.stderr[-3000:])
