import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ctcdecode_amd
B,T,V,K=64,500,10000,100
torch.manual_seed(7)
lp = torch.randn((B,T,V)).log_softmax(-1).cuda()
dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=40, cutoff_prob=0.99, beam_width=K, log_probs_input=True)
for _ in range(2): dec.decode_device(lp)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(3): dec.decode_device(lp, check=False)
torch.cuda.synchronize(); print("ms/batch", (time.perf_counter()-t0)/3*1e3)
