"""Generates tests/golden/frames_golden.json by running the reference's own loader
(data/video_pretrain_dataset.py::pretrain_video_dataset._load_video_from_path_decord) in THIS container on a fake
VideoReader whose "frames" are their own indices.  The reference module needs decord / av / torchvision, which
are absent here and carry no arithmetic on this path: they are registered as empty stub modules before the
file is loaded (the sampling code itself runs unmodified).  Random strategies are seeded per case through the
global streams the reference draws from (random.seed, numpy.random.seed)."""
import importlib.util
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data/video_pretrain_dataset.py"


def load_reference_module():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    stub("av")
    dec = stub("decord", VideoReader=object)
    dec.bridge = types.SimpleNamespace(set_bridge=lambda *_: None)
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    stub("data")
    stub("data.utils", pre_caption_minimum=lambda c, *_: c, wait_for_file=lambda *_: None)
    spec = importlib.util.spec_from_file_location("ref_video_pretrain_dataset", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class FakeVideoReader:
    def __init__(self, path, width=None, height=None):
        self.n = int(path.split(":")[1])

    def __len__(self):
        return self.n

    def get_batch(self, indices):
        return torch.as_tensor(np.asarray(indices, dtype=np.int64))


def main():
    mod = load_reference_module()
    mod.VideoReader = FakeVideoReader
    ds = object.__new__(mod.pretrain_video_dataset)
    cases = []
    seed = 0
    for strategy in ("uniform", "nlvl_uniform", "nlvl_rand", "rand", "headtail"):
        for vlen in (8, 9, 17, 30, 100, 301, 1000):
            for num_frm in (4, 8, 16):
                for span in (None, (1.0, 3.0, 25), (0.0, 7.5, 30)):
                    if num_frm > vlen or (span and strategy in ("rand", "headtail") and vlen != 100):
                        continue
                    seed += 1
                    st, et, fps = span if span else (None, None, -1)
                    ds.config = dict(frm_sampling_strategy=strategy, num_frm_train=num_frm, height=None, width=None,
                                     start_time=st, end_time=et, fps=fps)
                    random.seed(seed)
                    np.random.seed(seed)
                    got = ds._load_video_from_path_decord(f"fake:{vlen}")
                    # None: the reference caught an exception of its own sampling code (an empty span) and skips the video
                    cases.append(dict(strategy=strategy, vlen=vlen, num_frm=num_frm, start_time=st, end_time=et, fps=fps,
                                      seed=seed, indices=None if got is None else [int(i) for i in got]))
    json.dump(cases, open(os.path.join(HERE, "frames_golden.json"), "w"))
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
