import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ctcdecode_amd
from ctcdecode_amd import _native as n
torch.manual_seed(1234)
lp = torch.randn((256, 1000, 29)).log_softmax(-1).cuda()
dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(29)], cutoff_top_n=29, beam_width=100, log_probs_input=True)
dec.set_timing(True)
n.check(n.lib.ctcd_debug_set_profile(dec._handle, 1))
for _ in range(2):
    dec.decode_device(lp)
torch.cuda.synchronize()
ms = dec.last_kernel_ms()
prof = np.zeros((256, 16), np.int64)
n.check(n.lib.ctcd_debug_get_profile(dec._handle, prof.ctypes.data, 256))
tot = prof.sum(1).astype(np.float64)
rep = prof[:, 6].astype(np.float64)
print("instrumented kernel %.3f ms; per-workgroup total ticks: min %.3f mean %.3f max 1.000 (of max)" % (ms, tot.min() / tot.max(), tot.mean() / tot.max()))
print("exact-replay share per workgroup: min %.1f%% mean %.1f%% max %.1f%%; of the slowest workgroup %.1f%%" % (100 * (rep / tot).min(), 100 * (rep / tot).mean(), 100 * (rep / tot).max(), 100 * rep[tot.argmax()] / tot.max()))
print("total minus replay: min %.3f mean %.3f max %.3f (of max total)" % ((tot - rep).min() / tot.max(), (tot - rep).mean() / tot.max(), (tot - rep).max() / tot.max()))
NAMES = ["A", "B1", "B wait", "D compaction", "finish sort", "C3 (+tie resolve)", "D' exact replay", "E emit", "E update", "E fence", "frame load", "finish copy", "loop tail", "C1", "C2", "finish labels"]
order = np.argsort(tot)
slow = prof[order[-8:]].mean(0).astype(np.float64)
avg = prof.mean(0).astype(np.float64)
print("phase: mean workgroup -> mean of the 8 slowest (share of the slowest workgroup's total)")
for i in np.argsort(-(slow - avg)):
    print("  %-20s %6.2f%% -> %6.2f%%   (+%.2f%%)" % (NAMES[i], 100 * avg[i] / tot.max(), 100 * slow[i] / tot.max(), 100 * (slow[i] - avg[i]) / tot.max()))
