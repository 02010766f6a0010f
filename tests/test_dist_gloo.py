"""The N>1 path on CPU: world_size-2 Gloo processes shard a video list in balanced contiguous
blocks, gather their JSON results to rank 0, and the merged files equal the single-process
output (order included) — the property the 1/2/4/8-GPU runs rely on."""
import json
import os
import socket
import subprocess
import sys

from common import ROOT

WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
from vidil_amd import dist as vdist, capfilt, visual_tokenization as vt

rank, world, _ = vdist.init_distributed_mode(backend="gloo")
videos = [f"video{{i}}" for i in range(11)]
s, e = vdist.shard_bounds(len(videos))
items = []
for v in videos[s:e]:
    n = int(v[5:])
    items.append(dict(video_id=v, text=[] if n % 4 == 3 else [f"cap {{n}} é"], unfiltered_text=[f"cap {{n}} é", "x"]))
f, u = capfilt.collect_outputs(items)
capfilt.write_outputs({out!r}, f, u)
toks = {{v: dict(frame_tokens=[dict(objects=["o"])], caption=[], aggregated_tokens=dict(objects=["o"])) for v in videos[s:e]}}
vt.write_outputs({out!r}, toks)
t = vdist.max_over_ranks(float(rank + 1))
assert t == float(world), t
assert vdist.ranks_seen() == world, vdist.ranks_seen()          # one identity per rank (CPU ranks: host + pid)
# the collective of gather_json is chosen before it is issued: both forms deliver the same list to rank 0, None elsewhere
assert vdist._gather_supported()
a = vdist.gather_json(dict(rank=rank))
os.environ["VIDIL_GATHER"] = "allgather"
assert not vdist._gather_supported()
b = vdist.gather_json(dict(rank=rank))
assert a == b == ([dict(rank=r) for r in range(world)] if rank == 0 else None), (a, b)
vdist.barrier()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_RENDEZVOUS_ERRORS = ("Address already in use", "Connection refused", "Connection reset", "connect() timed out",
                      "failed to connect", "EADDRINUSE")


def _run(world, out):
    script = WORKER.format(root=ROOT, out=out)
    for attempt in range(3):
        # the port is free when probed, not necessarily a moment later: a lost race for it is a rendezvous failure of
        # the test harness, retried on a fresh port; any other failure of a worker fails the test at once
        port = _free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [(p, p.communicate(timeout=240)[0].decode()) for p in procs]
        bad = [o for p, o in outs if p.returncode != 0]
        if not bad:
            return
        if attempt == 2 or not any(e in o for o in bad for e in _RENDEZVOUS_ERRORS):
            raise AssertionError("\n".join(bad))
        print("rendezvous failure, retrying on a new port:\n" + "\n".join(bad), file=sys.stderr)


def test_two_rank_gather_equals_single_process(tmp_path):
    out1, out2 = str(tmp_path / "w1"), str(tmp_path / "w2")
    _run(1, out1)
    _run(2, out2)
    for name in ("video_text_CapFilt.json", "video_text_Cap.json", "visual_tokens.json"):
        a = open(os.path.join(out1, name)).read()
        b = open(os.path.join(out2, name)).read()
        assert a == b, name
    d = json.load(open(os.path.join(out2, "video_text_CapFilt.json")))
    assert list(d.keys()) == [f"video{i}" for i in range(11) if i % 4 != 3]      # filtered-out videos drop, order kept
    assert list(json.load(open(os.path.join(out2, "video_text_Cap.json"))).keys()) == [f"video{i}" for i in range(11)]


def test_gather_collective_is_decided_before_it_is_issued():
    """no try/except around the collective — the source of gather_json holds no exception handler, and an error
    inside the collective propagates (here: a process group whose gather raises)."""
    import ast
    import inspect

    from vidil_amd import dist as vdist

    tree = ast.parse(inspect.getsource(vdist.gather_json))
    assert not [n for n in ast.walk(tree) if isinstance(n, ast.Try)]
    assert vdist.ranks_seen() == 1 and vdist.gather_json({"a": 1}) == [{"a": 1}]      # no process group: identity
