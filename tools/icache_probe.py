#!/usr/bin/env python3
"""Workload for instruction-cache counters of the scorer's kernels (the run-time-layout / wide-beam / hook kernels
with a scorer are 81-98 KB, over the 64 KB instruction cache two CUs share -- does it cost them anything?).
    python tools/icache_probe.py <variant>        variant: lm2 (word-model kernel, fixed layout, 49 KB) | lm1 (general LM kernel, fixed layout, 55 KB) |
                                                   rt (the same decode on the run-time layout, 84 KB) | wide (beam 300: first wide-beam layout, 94 KB) | nolm (north-star kernel, 40 KB)
Run under rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH (tools/icache_probe.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variant = sys.argv[1]
if variant == "lm1":
    os.environ["CTCD_GENERAL_LM_KERNEL"] = "1"
import torch  # noqa: E402

import ctcdecode_amd  # noqa: E402

labels = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
g = torch.Generator(device="cpu").manual_seed(7)
B, T, K = (128, 1500, 100) if variant != "wide" else (64, 400, 300)
lp = torch.randn((B, T, 29), generator=g).log_softmax(-1).cuda()
kw = {} if variant == "nolm" else dict(model_path=os.path.join(ROOT, "tests", "data", "test.arpa"), alpha=0.5, beta=1.0)
dec = ctcdecode_amd.CTCBeamDecoder(labels, cutoff_top_n=29, beam_width=K, log_probs_input=True, **kw)
dec.set_timing(True)
if variant == "rt":
    dec.set_fixed_layout(False)
for _ in range(3):
    dec.decode_device(lp)
torch.cuda.synchronize()
print("%s kernel_ms %.3f" % (variant, dec.last_kernel_ms()))
