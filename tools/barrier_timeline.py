#!/usr/bin/env python3
"""Per-wave timeline of a few frames of one utterance (timeline build of the kernel, ctcd_debug_timeline): the shader
clock every wave stamps when it arrives at / leaves each workgroup barrier and at a few extra points of a frame.
Prints, per stamp, the clocks until the next stamp for every wave (averaged over the recorded frames): the rows between
a "leave" and the next "arrive" are work, the rows between an "arrive" and its "leave" are waiting for the slowest wave.

    python tools/barrier_timeline.py [--frame0 500 --frames 6 --threads 0] [--out profiles/x_timeline.json]
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# stamps of one frame on the common path (beam_core.h: x.tick() calls and the barriers of step())
LABELS = ["loop top (row prefetch issued) -> step prologue", "A1 subtree ends + painting", "(tick -> barrier)", "wait: A1 barrier",
          "A2 slot offsets, existing children", "wait: A2 barrier", "B score candidates + histogram", "wait: B barrier",
          "C1 find bucket (wave 0)", "wait: C1 barrier", "C2 list bucket + survivor bitmap", "wait: C2 barrier",
          "C3 rank in bucket (wave 0)", "wait: C3 barrier", "D expand bitmap (wave 0)", "wait: D barrier",
          "E emit next beam (+ resets on idle waves)", "wait: end-of-frame barrier", "state update", "loop back-edge"]
# ... on the path of the speculative select (round 4): the histogram select's four stages are two
LABELS_SPEC = LABELS[:6] + ["B score candidates + hot list", "wait: B barrier", "R1 rank the hot list (all waves)", "wait: R1 barrier",
                            "R2 survivors in slot order (all waves)", "wait: R2 barrier"] + LABELS[16:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--V", type=int, default=29)
    ap.add_argument("--beam", type=int, default=100)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--frame0", type=int, default=500)
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--out", default="")
    ap.add_argument("--repeat", type=int, default=1, help="launches to average over (frame0 moves by 37 each)")
    ap.add_argument("--lm", default="", help="ARPA model: time the LM tier's kernel (labels _ ' space a..z)")
    ap.add_argument("--all-paths", action="store_true", help="print the rows of every group of frames (by stamps per frame), not only the most frequent one")
    ap.add_argument("--kind", default="randn", help="randn | blank (+6 on the blank logit: nearly every frame takes the speculative select)")
    a = ap.parse_args()
    import numpy as np
    import torch

    import ctcdecode_amd
    from ctcdecode_amd import _native

    g = torch.Generator(device="cpu").manual_seed(1234)
    lg = torch.randn((a.batch, a.T, a.V), generator=g)
    if a.kind == "blank":
        lg[:, :, 0] += 6.0
    lp = lg.log_softmax(-1).cuda()
    labels = [str(i) for i in range(a.V)]
    kw = {}
    if a.lm:
        labels = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
        kw = dict(model_path=a.lm, alpha=0.5, beta=1.0)
    dec = ctcdecode_amd.CTCBeamDecoder(labels, cutoff_top_n=a.V, beam_width=a.beam, log_probs_input=True, **kw)
    if a.threads:
        dec.set_threads(a.threads)
    dec.set_timing(True)
    _native.check(_native.lib.ctcd_debug_set_profile(dec._handle, 1))
    MARK = 1 << 62
    groups = {}  # stamps per frame -> list of [waves, stamps] arrays of clock differences (marker -> stamp 0 -> ... -> next marker)
    frame_clocks = []
    kernel_ms = 0.0
    overflowed = 0
    for rep in range(a.repeat):  # (the LM build keeps 64 stamps per wave: average over several launches)
        _native.check(_native.lib.ctcd_debug_timeline(dec._handle, a.frame0 + 37 * rep, a.frames, None))
        dec.decode_device(lp)
        torch.cuda.synchronize()
        dec.decode_device(lp)
        torch.cuda.synchronize()
        kernel_ms = dec.last_kernel_ms()
        cap = _native.lib.ctcd_debug_timeline_cap()
        buf = np.zeros((16, cap), np.int64)
        _native.check(_native.lib.ctcd_debug_timeline(dec._handle, 0, 0, buf.ctypes.data_as(ctypes.c_void_p)))
        if a.lm or a.beam > 128:  # the LM build of the timeline kernel records half as many stamps per wave, the wide-beam build too
            cap //= 2
            buf = buf.reshape(-1)[:16 * cap].reshape(16, cap)
        nw = int((buf[:, 0] != 0).sum())
        n = int((buf[0] != 0).sum())
        if n >= cap:  # (a frame with an exact replay stamps every barrier of it: such launches are counted, not shown)
            overflowed += 1
            continue
        marks = [i for i in range(n) if buf[0, i] & MARK]
        t = (buf[:nw, :n] & (MARK - 1)).astype(np.float64)
        for m0, m1 in zip(marks[:-1], marks[1:]):  # whole frames: marker to marker (every wave records the same sequence)
            d = np.diff(t[:, m0:m1 + 1], axis=1)
            groups.setdefault(m1 - m0 - 1, []).append(d)
            frame_clocks.append(float(t[0, m1] - t[0, m0]))
    per = max(groups, key=lambda k: len(groups[k]))
    acc = np.mean(groups[per], axis=0)
    cnt = len(groups[per])
    clocks_per_frame = float(np.mean([g.sum(axis=1)[0] for g in groups[per]]))
    if overflowed:
        print("%d of %d launches filled the timeline buffer (more stamps than it holds: fewer --frames) and are left out" % (overflowed, a.repeat))
    print("timeline build: kernel %.3f ms; %d waves; frames recorded by stamps per frame: %s; showing the %d-stamp frames (%d of them), %.0f clocks each"
          % (kernel_ms, nw, {k: len(v) for k, v in sorted(groups.items())}, per, cnt, clocks_per_frame))
    # (row 0 = marker -> first stamp: the loop top; the marker itself costs what a stamp costs)
    labels = (["(frame marker -> loop top)"] + LABELS) if per == len(LABELS) else (["(frame marker -> loop top)"] + LABELS_SPEC) if per == len(LABELS_SPEC) else None
    overhead = float(acc[:, 3].mean()) if labels else 0.0
    print("one stamp costs about %.0f clocks (the row '(tick -> barrier)' has nothing else in it)" % overhead)
    rows = []
    for i in range(acc.shape[1]):
        lab = labels[i] if labels else "stamp %d" % i
        rows.append({"stamp": i, "what": lab, "max": float(acc[:, i].max()), "median": float(np.median(acc[:, i])), "min": float(acc[:, i].min()),
                     "per_wave": [float(v) for v in acc[:, i]]})
        print("%2d %-52s max %5.0f  med %5.0f  min %5.0f | %s" % (i, lab[:52], acc[:, i].max(), np.median(acc[:, i]), acc[:, i].min(),
                                                                 " ".join("%4.0f" % v for v in acc[:, i])))
    if a.all_paths:
        for k, v in sorted(groups.items()):
            if k == per:
                continue
            ac = np.mean(v, axis=0)
            print("-- frames with %d stamps (%d of them): slowest wave / median wave per interval" % (k, len(v)))
            for i in range(ac.shape[1]):
                print("   %2d  max %5.0f  med %5.0f" % (i, ac[:, i].max(), np.median(ac[:, i])))
    per_group = {str(k): float(np.mean([g.sum(axis=1)[0] for g in v])) for k, v in groups.items()}
    print("clocks per frame by path (stamps per frame -> clocks, stamps included):", per_group)
    if a.out:
        json.dump({"kernel_ms_timeline_build": kernel_ms, "waves": nw, "stamps_per_frame": per, "clocks_per_frame": clocks_per_frame,
                   "stamp_overhead_clocks": overhead, "frames_averaged": cnt, "frame0": a.frame0, "rows": rows,
                   "frames_by_stamps": {str(k): len(v) for k, v in groups.items()}, "clocks_per_frame_by_stamps": per_group}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
