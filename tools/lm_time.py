import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ctcdecode_amd
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
B, T, V, K = 128, 1500, 29, 100
labels = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
torch.manual_seed(3)
lp = torch.randn((B, T, V)).log_softmax(-1).cuda()
dec = ctcdecode_amd.CTCBeamDecoder(labels, beam_width=K, log_probs_input=True, model_path=os.path.join(ROOT, "tests", "data", "test.arpa"), alpha=0.5, beta=1.0)
dec.set_timing(True)
ms = []
for _ in range(8):
    r = dec.decode_device(lp)
    torch.cuda.synchronize()
    ms.append(dec.last_kernel_ms())
ms = sorted(ms[2:])
print("LM kernel ms: min %.3f median %.3f  lens %d" % (ms[0], ms[len(ms)//2], int(r[3].sum())))
