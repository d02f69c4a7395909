import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, ctcdecode_amd
from ctcdecode_amd import _native
B,T,V,K=256,1000,29,100
g = torch.Generator(device="cpu").manual_seed(1234)
lp = torch.randn((B,T,V), generator=g).log_softmax(-1).cuda()
dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=V, beam_width=K, log_probs_input=True)
_native.check(_native.lib.ctcd_debug_set_profile(dec._handle, 1))
NF=6; REC=int(sys.argv[1]) if len(sys.argv)>1 else 20
_native.check(_native.lib.ctcd_debug_timeline(dec._handle, 500, NF, None))
dec.set_timing(True)
dec.decode_device(lp); torch.cuda.synchronize()
dec.decode_device(lp); torch.cuda.synchronize()
print('timeline build kernel ms', dec.last_kernel_ms())
cap=_native.lib.ctcd_debug_timeline_cap()
buf=np.zeros((16,cap),np.int64)
_native.check(_native.lib.ctcd_debug_timeline(dec._handle,0,0,buf.ctypes.data_as(ctypes.c_void_p)))
n=int((buf[0]!=0).sum()); print('clocks first->last record', buf[0,n-1]-buf[0,0]); print("records per wave", n, "per frame", n/NF)
t=buf[:, :n].astype(np.float64)
d=np.diff(t,axis=1)   # [16, n-1]
per=n//NF
# average over frames 1..NF-1 (skip first)
acc=np.zeros((16,per)); cnt=0
for f in range(1,NF-1):
    acc+=d[:, f*per:(f+1)*per]; cnt+=1
acc/=cnt
np.set_printoptions(linewidth=250, suppress=True)
print("rows = record index within frame (delta to the next record); columns = waves 0..15")
for i in range(per):
    print("%2d"%i, " ".join("%5.0f"%x for x in acc[:,i]), "  max %5.0f med %5.0f"%(acc[:,i].max(), np.median(acc[:,i])))
