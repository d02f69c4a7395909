#!/usr/bin/env python3
"""One launch each of the kernels bench.py's headline does not touch, for rocprofv3 (kernel stats / FETCH_SIZE / WRITE_SIZE):
the LM instantiation of the decode kernel (configs[4] per-GPU shape, tests/data/test.arpa), the raw-logit kernels at configs[3]'s B=64, T=500,
V=10000 (prune_logits_wg_kernel, log_softmax_rows_wg_kernel, and the one-wave log_softmax_rows_kernel + separate prune they replace), expand_compact_kernel (one 256-utterance configs[1] batch), the two-workgroups-per-CU
build (512 utterances), and -- as the calibration of the HBM byte counters MI355X_MICROARCH.md asks for -- a float4 copy of
exactly 1 GiB (torch's vectorised copy kernel).  Prints one JSON line with HIP-event / wall timings.
    python tools/profile_extras.py [--only lm,softmax,expand,occ2,copy]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="lm,softmax,expand,occ2,copy")
    a = ap.parse_args()
    want = set(a.only.split(","))
    import torch

    import ctcdecode_amd

    dev = torch.device("cuda", 0)
    out = {}
    g = torch.Generator(device="cpu").manual_seed(7)

    def timed(fn, reps=2):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts)

    if "lm" in want:
        labels = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
        lp = torch.randn((128, 1500, 29), generator=g).log_softmax(-1).to(dev)
        dec = ctcdecode_amd.CTCBeamDecoder(labels, model_path=os.path.join(ROOT, "tests", "data", "test.arpa"), alpha=0.5, beta=1.0, cutoff_top_n=40,
                                           beam_width=100, log_probs_input=True, device=dev)
        dec.set_timing(True)
        ms = timed(lambda: dec.decode_device(lp, None))
        out["lm_configs4_shape"] = {"wall_ms": round(ms, 3), "kernel_ms": round(dec.last_kernel_ms(), 3), "bytes_in": lp.numel() * 4}
    if "softmax" in want:
        x = torch.randn((64, 500, 10000), generator=g).to(dev)
        dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(10000)], cutoff_top_n=40, cutoff_prob=0.99, beam_width=100, logits_input=True, device=dev)
        dec.set_timing(True)
        ms = timed(lambda: dec.decode_device(x, None))
        # (round 6: in front of a prune the logits go through ONE kernel, prune_logits_wg_kernel; the stand-alone normalisation --
        #  log_softmax_rows_wg_kernel -- and the two-pass form it replaces are launched as well so that all of them show up in the trace)
        out["logits_configs3_shape"] = {"wall_ms": round(ms, 3), "fused_prune_kernel_ms": round(dec.last_prune_ms(), 3), "logits_bytes_in": x.numel() * 4}
        y = dec.log_softmax(x)
        ms = timed(lambda: dec.log_softmax(x))
        out["log_softmax_configs3_rows"] = {"wall_ms_incl_output_alloc": round(ms, 3), "softmax_bytes_in_plus_out": 2 * x.numel() * 4}
        dec.set_fused_logits(False)
        ms = timed(lambda: dec.decode_device(x, None))
        out["logits_configs3_shape_two_pass_one_wave_kernels"] = {"wall_ms": round(ms, 3), "separate_prune_kernel_ms": round(dec.last_prune_ms(), 3)}
        del y
    if "expand" in want:
        lp = torch.randn((256, 1000, 29), generator=g).log_softmax(-1).to(dev)
        dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(29)], cutoff_top_n=29, beam_width=100, log_probs_input=True, device=dev)
        hdr, ent, lab, sc, ln = dec.decode_compact(lp, None)
        ms = timed(lambda: dec.expand_compact(hdr, ent, lab, 1000))
        out["expand_compact_256x100x1000"] = {"wall_ms": round(ms, 3), "bytes_written": 2 * 256 * 100 * 1000 * 4, "compact_labels": int(lab.numel())}
    if "occ2" in want:
        lp = torch.randn((512, 1000, 29), generator=g).log_softmax(-1).to(dev)
        dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(29)], cutoff_top_n=29, beam_width=100, log_probs_input=True, device=dev)
        dec.set_timing(True)
        ms = timed(lambda: dec.decode_device(lp, None))
        out["two_workgroups_per_cu_512_utterances"] = {"wall_ms": round(ms, 3), "kernel_ms": round(dec.last_kernel_ms(), 3)}
    if "copy" in want:
        n = 1 << 28  # 2^28 float32 = 1 GiB
        src = torch.ones((n,), dtype=torch.float32, device=dev)
        dst = torch.empty_like(src)
        ms = timed(lambda: dst.copy_(src), reps=3)
        out["float4_copy_1GiB"] = {"wall_ms": round(ms, 3), "bytes_read": n * 4, "bytes_written": n * 4, "GBps_read_plus_write": round(2 * n * 4 / ms / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
