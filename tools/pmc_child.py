#!/usr/bin/env python3
"""The workload bench.py's counter passes profile (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around this script): the headline
batch decoded a few times, nothing else -- the passes must be cheap enough to run inside every bench.py run.
    python tools/pmc_child.py B T V beam threads [seed]        (seed: bench.py passes its rank-0 seed, 1234)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctcdecode_amd  # noqa: E402

B, T, V, K, threads = (int(v) for v in sys.argv[1:6])
seed = int(sys.argv[6]) if len(sys.argv) > 6 else 1234
g = torch.Generator(device="cpu").manual_seed(seed)  # (bench.py main(): the same expression -- the same batch)
lp = torch.randn((max(B, 1), T, V), generator=g, dtype=torch.float32).log_softmax(-1)[:B].cuda()
dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=V, beam_width=K, blank_id=0, log_probs_input=True)
if threads:
    dec.set_threads(threads)
for _ in range(4):
    res = dec.decode_device(lp, None, check=False)
torch.cuda.synchronize()
