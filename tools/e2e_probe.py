#!/usr/bin/env python3
"""Where the time of the drop-in decode() goes (configs[1]): input side, kernel, result delivery, per host-thread count."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctcdecode_amd

B, T, V, K = 256, 1000, 29, 100
torch.manual_seed(1)
lp_cpu = torch.randn((B, T, V)).log_softmax(-1)
lp_pin = lp_cpu.pin_memory()
lp_dev = lp_cpu.cuda()


def med(f, reps=7, warm=2):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


for nproc in (1, 4, 8, 16, 32, 64):
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], beam_width=K, log_probs_input=True, num_processes=nproc)
    print("num_processes=%2d  decode(cpu pageable in) %.2f ms | decode(pinned in) %.2f ms | decode(device in) %.2f ms" % (
        nproc, med(lambda: dec.decode(lp_cpu)), med(lambda: dec.decode(lp_pin)), med(lambda: dec.decode(lp_dev))), flush=True)
dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], beam_width=K, log_probs_input=True)
print("decode_device (HBM in/out) %.2f ms | decode_compact %.2f ms | decode_padded(cpu in) %.2f ms | h2d pageable %.2f ms | h2d pinned %.2f ms" % (
    med(lambda: dec.decode_device(lp_dev)), med(lambda: dec.decode_compact(lp_dev)), med(lambda: dec.decode_padded(lp_cpu)),
    med(lambda: lp_cpu.cuda()), med(lambda: lp_pin.cuda())))
hdr, ent, labels, sc, ln = dec.decode_compact(lp_dev)
print("compact labels: %d (%.1f MB) vs padded 2x[B,K,T] int32 = %.1f MB" % (labels.numel(), labels.numel() * 4 / 1e6, 2 * B * K * T * 4 / 1e6))
print("expand_compact on device %.3f ms" % med(lambda: dec.expand_compact(hdr, ent, labels, T)))
