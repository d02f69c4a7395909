#!/usr/bin/env python3
"""The scorer hook's cost (bench.py time_scorer_hook) and the two-launches-in-flight rule (bench.py time_inflight) on their own.
    python tools/hook_probe.py [--big] [--inflight] [--threads=1 --threads=4 ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import ctcdecode_amd  # noqa: E402

dev = torch.device("cuda", 0)
labels = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
arpa = os.path.join(ROOT, "tests", "data", "test.arpa")
if "--inflight" in sys.argv:
    lp128 = bench.synth_rows(torch, 128, 1500, 29, 7).to(dev)
    for name, kw in (("no LM", {}), ("test.arpa", dict(model_path=arpa, alpha=0.5, beta=1.0))):
        one = bench.time_inflight(torch, ctcdecode_amd, dev, lp128, labels, 100, 1, **kw)
        two = bench.time_inflight(torch, ctcdecode_amd, dev, lp128, labels, 100, 2, **kw)
        print("128 utterances x 1500 frames, %s: one in flight %.3f ms/batch (%.0f utt/s), two in flight %.3f ms/batch (%.0f utt/s)" % (name, one * 1e3, 128 / one, two * 1e3, 128 / two))
    lp256 = bench.synth_rows(torch, 256, 1000, 29, 1234).to(dev)
    for k in (1, 2, 3):
        dt = bench.time_inflight(torch, ctcdecode_amd, dev, lp256, [str(i) for i in range(29)], 100, k, steps=20)
        print("headline batch, default build, %d in flight: %.3f ms/batch (%.0f utt/s)" % (k, dt * 1e3, 256 / dt))
threads = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--threads=")] or [1]
for th in threads:
    print(json.dumps(bench.time_scorer_hook(torch, ctcdecode_amd, dev, arpa, labels, threads=th), indent=1))
    print(json.dumps(bench.time_scorer_hook(torch, ctcdecode_amd, dev, arpa, labels, transcripts=True, threads=th), indent=1))
if "--big" in sys.argv:
    import importlib.util
    import tempfile

    spec = importlib.util.spec_from_file_location("make_big_lm", os.path.join(ROOT, "tools", "make_big_lm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    big = os.path.join(tempfile.gettempdir(), "ctcd_big_words_50k.arpa")
    if not os.path.exists(big):
        mod.make(big)
    for th in threads:
        print(json.dumps(bench.time_scorer_hook(torch, ctcdecode_amd, dev, big, labels, transcripts=True, threads=th), indent=1))
