#!/usr/bin/env python3
"""Duration of the vocabulary-prune pass alone at the configs[3] shape (B=64, T=500, V=10000, top_n 40, cutoff_prob 0.99);
--logits: the rows are raw logits (logits_input=True: the fused logits -> candidates pass)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctcdecode_amd
B, T, V, K = 64, 500, 10000, 100
g = torch.Generator(device="cpu").manual_seed(7)
lp = torch.randn((B, T, V), generator=g).log_softmax(-1).cuda()
LOGITS = "--logits" in sys.argv
dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=40, cutoff_prob=0.99, beam_width=K, log_probs_input=True, **({"logits_input": True} if LOGITS else {}))
dec.set_timing(True)
ms = []
for _ in range(5):
    dec.decode_device(lp)
    torch.cuda.synchronize()
    ms.append(dec.last_prune_ms())
m = min(ms[1:])
print("%s %s prune %.3f ms = %.0f GB/s" % (os.path.basename(os.environ.get("CTCDECODE_AMD_LIB", "default")), "logits" if LOGITS else "log-probs", m, B * T * V * 4 / m / 1e6))
