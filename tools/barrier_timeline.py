#!/usr/bin/env python3
"""Per-wave barrier timeline of a few frames of one utterance (instrumented kernel build): when every wave arrives at
and leaves each workgroup barrier, in shader clocks.  Shows, per barrier interval, how long the slowest wave worked and
which wave it was -- the critical path of a frame.

    python tools/barrier_timeline.py [--frame0 500 --frames 4 --threads 0] [--out profiles/x_timeline.json]
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--V", type=int, default=29)
    ap.add_argument("--beam", type=int, default=100)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--frame0", type=int, default=500)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import numpy as np
    import torch

    import ctcdecode_amd
    from ctcdecode_amd import _native

    g = torch.Generator(device="cpu").manual_seed(1234)
    lp = torch.randn((a.batch, a.T, a.V), generator=g).log_softmax(-1).cuda()
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(a.V)], cutoff_top_n=a.V, beam_width=a.beam, log_probs_input=True)
    if a.threads:
        dec.set_threads(a.threads)
    _native.check(_native.lib.ctcd_debug_set_profile(dec._handle, 1))
    _native.check(_native.lib.ctcd_debug_timeline(dec._handle, a.frame0, a.frames, None))
    dec.decode_device(lp)
    torch.cuda.synchronize()
    cap = _native.lib.ctcd_debug_timeline_cap()
    buf = np.zeros((16, cap), np.int64)
    _native.check(_native.lib.ctcd_debug_timeline(dec._handle, 0, 0, buf.ctypes.data_as(ctypes.c_void_p)))
    nw = int((buf[:, 0] != 0).sum())
    n = int((buf[0] != 0).sum())
    t = buf[:nw, :n].astype(np.float64)
    arrive, leave = t[:, 0::2], t[:, 1::2]
    nb = min(arrive.shape[1], leave.shape[1])
    arrive, leave = arrive[:, :nb], leave[:, :nb]
    per_frame = nb // a.frames
    rows = []
    # interval i: from the release of barrier i-1 (max over waves of `leave`) to the last arrival at barrier i
    for i in range(1, nb):
        start = leave[:, i - 1].min()
        work = arrive[:, i] - leave[:, i - 1]          # per wave: time between leaving the previous barrier and arriving here
        rows.append({"barrier": i, "in_frame": i % per_frame if per_frame else i, "span": float(arrive[:, i].max() - start),
                     "slowest_wave": int(work.argmax()), "slowest_work": float(work.max()), "median_work": float(np.median(work)),
                     "barrier_latency": float(leave[:, i].min() - arrive[:, i].max()), "work_per_wave": [float(x) for x in work]})
    total = float(arrive[:, nb - 1].max() - leave[:, 0].min())
    res = {"waves": nw, "barriers_recorded": nb, "barriers_per_frame": per_frame, "frames": a.frames, "clocks_total": total,
           "clocks_per_frame": total / max(1, a.frames - 1.0 / max(per_frame, 1)), "intervals": rows}
    print("waves %d, %d barriers (%d per frame); %.0f clocks per frame" % (nw, nb, per_frame, res["clocks_per_frame"]))
    # average the intervals with the same position in the frame
    if per_frame:
        print("%-4s %8s %8s %8s %8s  %s" % ("pos", "span", "slowest", "median", "barrier", "slowest wave (per frame)"))
        for pos in range(per_frame):
            sel = [r for r in rows if r["in_frame"] == pos]
            if not sel:
                continue
            print("%-4d %8.0f %8.0f %8.0f %8.0f  %s" % (pos, np.mean([r["span"] for r in sel]), np.mean([r["slowest_work"] for r in sel]),
                                                   np.mean([r["median_work"] for r in sel]), np.mean([r["barrier_latency"] for r in sel]),
                                                   [r["slowest_wave"] for r in sel]))
    if a.out:
        json.dump(res, open(a.out, "w"))


if __name__ == "__main__":
    main()
