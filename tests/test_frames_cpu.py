"""vidil_amd.frames.frame_indices against vectors produced by the reference's own loader
(tests/golden/make_frames_golden.py ran data/video_pretrain_dataset.py on a fake VideoReader)."""
import json
import os
import random

import numpy as np
import pytest

from vidil_amd.frames import STRATEGIES, clip_range, frame_indices, kmeans_pick

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "frames_golden.json")))


def _run(c, **kw):
    return frame_indices(c["vlen"], c["num_frm"], c["strategy"], start_time=c["start_time"], end_time=c["end_time"],
                         fps=c["fps"], **kw)


def test_every_reference_case_global_streams():
    assert {c["strategy"] for c in CASES} == set(STRATEGIES) - {"clip-kmeans"}
    for c in CASES:
        random.seed(c["seed"])
        np.random.seed(c["seed"])
        if c["indices"] is None:                       # the reference's own code raised (empty span): so do we
            with pytest.raises(Exception):
                _run(c)
            continue
        got = _run(c)
        assert got == c["indices"], c
        assert all(isinstance(i, int) for i in got)


def test_injected_streams_match_the_global_ones():
    for c in CASES:
        if c["indices"] is None or c["strategy"] not in ("nlvl_rand", "rand", "headtail"):
            continue
        got = _run(c, py_random=random.Random(c["seed"]), np_random=np.random.RandomState(c["seed"]))
        assert got == c["indices"], c


def test_span_and_errors():
    assert clip_range(100) == (0, 100)
    assert clip_range(100, 1.0, 3.0, 25) == (25, 75)
    assert clip_range(50, 1.0, 3.0, 25) == (25, 50)
    with pytest.raises(AssertionError):
        clip_range(100, 1.0, 3.0, -1)
    with pytest.raises(NotImplementedError):
        frame_indices(10, 4, "middle")
    with pytest.raises(ValueError):
        frame_indices(10, 4, "clip-kmeans")
    with pytest.raises(ValueError):                    # random.sample: more frames than the video has
        frame_indices(3, 8, "rand")
    assert frame_indices(40, 4, "clip-kmeans", clip_select=lambda n: [3, 9, 20, 31][:n]) == [3, 9, 20, 31]


def test_kmeans_pick_on_separated_clusters():
    rng = np.random.RandomState(0)
    centers = np.array([[10.0, 0, 0], [0, 10.0, 0], [0, 0, 10.0], [-10.0, -10.0, 0]])
    member = np.array([0, 0, 1, 1, 1, 2, 3, 3, 2, 0, 1, 3])
    emb = centers[member] + 0.01 * rng.randn(len(member), 3)
    cand = np.arange(0, 2 * len(member), 2)
    picked = kmeans_pick(emb, 4, cand, np_random=np.random.RandomState(5))
    assert picked == sorted(picked) and len(picked) == 4
    assert sorted(member[np.array(picked) // 2].tolist()) == [0, 1, 2, 3]        # one frame from every scene
    assert picked == kmeans_pick(emb, 4, cand, np_random=np.random.RandomState(5))
