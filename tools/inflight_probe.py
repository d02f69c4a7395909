import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ctcdecode_amd
B, T, V, K = 256, 1000, 29, 100
g = torch.Generator(device="cpu").manual_seed(1234)
lp = torch.randn((B, T, V), generator=g).log_softmax(-1).cuda()
labels = [str(i) for i in range(V)]
for nfl in (1, 2, 3):
    decs = [ctcdecode_amd.CTCBeamDecoder(labels, cutoff_top_n=V, beam_width=K, log_probs_input=True) for _ in range(nfl)]
    streams = [torch.cuda.Stream() for _ in range(nfl)]
    def run(steps):
        for i in range(steps):
            with torch.cuda.stream(streams[i % nfl]):
                decs[i % nfl].decode_device(lp, None, check=False)
        torch.cuda.synchronize()
    run(6)
    t0 = time.perf_counter(); run(30); dt = time.perf_counter() - t0
    for d in decs:
        ctcdecode_amd._native.check(ctcdecode_amd._native.lib.ctcd_check_status(d._handle, B))
    print("launches in flight %d: %.3f ms per batch, %.0f utt/s" % (nfl, dt / 30 * 1e3, B * 30 / dt))
