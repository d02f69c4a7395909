#!/usr/bin/env python3
"""Kernel time of the LM tier against the same decode without a scorer (configs[1] shape; labels blank, ', space, a..z)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctcdecode_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T, V, K = 256, 1000, 29, 100
labels = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
torch.manual_seed(3)
lp = torch.randn((B, T, V)).log_softmax(-1).cuda()

def run(name, **kw):
    fixed = kw.pop("fixed", True)
    dec = ctcdecode_amd.CTCBeamDecoder(labels, beam_width=K, log_probs_input=True, **kw)
    if not fixed:
        dec.set_fixed_layout(False)
    dec.set_timing(True)
    ms = []
    for _ in range(4):
        r = dec.decode_device(lp)
        torch.cuda.synchronize()
        ms.append(dec.last_kernel_ms())
    print("%-60s kernel %.2f ms  (%.2f us/frame)  mean top length %.0f" % (name, min(ms[1:]), min(ms[1:]) * 1e3 / T, float(r[3][:, 0].float().mean())), flush=True)

run("no scorer, fixed layout (product default)")
run("no scorer, run-time layout", fixed=False)
for arpa in ("test.arpa", "abcd_words.arpa", "chars.arpa"):
    run("scorer %s alpha 0.5 beta 1.0" % arpa, model_path=os.path.join(ROOT, "tests", "data", arpa), alpha=0.5, beta=1.0)
