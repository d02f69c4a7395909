import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ctcdecode_amd
B,T,V,K=256,1000,29,100
torch.manual_seed(1)
lp_cpu = torch.randn((B,T,V)).log_softmax(-1)
dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], beam_width=K, log_probs_input=True)
for _ in range(2): r = dec.decode(lp_cpu)
t0=time.perf_counter()
for _ in range(5): r = dec.decode(lp_cpu)
dt=(time.perf_counter()-t0)/5
print("decode() host tensors in/out: %.2f ms per batch = %.0f utt/s" % (dt*1e3, B/dt))
lp = lp_cpu.cuda()
t0=time.perf_counter()
for _ in range(5): r = dec.decode(lp)
dt=(time.perf_counter()-t0)/5
print("decode() device input, host output: %.2f ms per batch = %.0f utt/s" % (dt*1e3, B/dt))
