#!/usr/bin/env python3
"""Streaming timings: python tools/stream_probe.py [B ...]  -- bench.py's time_streaming at several chunk sizes."""
import importlib.util
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import ctcdecode_amd

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
dev = torch.device("cuda", 0)
for B in [int(a) for a in sys.argv[1:]] or [256]:
    g = torch.Generator(device="cpu").manual_seed(1234)
    lp = torch.randn((B, 1000, 29), generator=g).log_softmax(-1).to(dev)
    for chunk in (50, 200, 1000):
        r = bench.time_streaming(torch, ctcdecode_amd, dev, lp, 29, 100, chunk=chunk, reps=2)
        r.pop("what")
        print(B, json.dumps(r))
