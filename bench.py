#!/usr/bin/env python3
"""bench.py -- utterances/s of CTC prefix beam-search decoding on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1], per GPU): B=256 utterances, T=1000 frames, V=29 labels, beam_width=100,
cutoff_top_n=29 (no pruning), no LM; inputs are float32 log-softmax of N(0,1) logits, resident in HBM before the
timed region.  A "step" = one decode of that whole batch through the product path (C ABI -> HIP kernel).  With N GPUs
every rank decodes its own B utterances (weak scaling, no data-path collective) and rank 0 then gathers the results
over RCCL (north_star's "trivial gather") inside the timed region; by default batch i's gather overlaps the decode of
batch i+1 (ctcdecode_amd.distributed.ResultGatherer), all gathers complete before the clock stops.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: this process launches the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

    python bench.py --config {1,2,4} ...     BASELINE.json's other multi-GPU configurations (default 1 = the headline):
        2 = configs[2]: B=2048 utterances, beam 500, T=2000, no LM -- STRONG scaling: the 2048 utterances are cut into
            contiguous blocks of ceil(2048 / N) (ctcdecode_amd.distributed.shard_bounds), one block per rank;
        4 = configs[4]: B=1024, beam 100, T=1500, with the LM scorer on tests/data/test.arpa (alpha 0.5, beta 1.0), strong scaling.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the definition of every field).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Critical-path floor of ONE frame of the decode kernel's phase structure, in shader clocks, from the primitives measured
# on this GPU (tools/ubench): dependent LDS round trip 68, workgroup barrier at 1024 threads 78,
# six-step DPP scan 100, LDS atomic 50, the dependent binary64 chain of one log_sum_exp ~300.
FRAME_FLOOR = {
    "A1 subtree ends + painting (2 LDS round trips, ballot search, atomics)": 250,
    "A2 slot offsets (1 round trip)": 150,
    "B score + histogram (1 round trip, log_sum_exp, atomic)": 420,
    "C1 find bucket (2 round trips, 2 DPP scans)": 350,
    "C2 list the bucket (1 round trip, atomic append)": 200,
    "C3 rank inside the bucket (1 round trip, compare)": 200,
    "D ordered compaction (1 round trip, DPP scan, scatter)": 250,
    "E emit the next beam (4 dependent round trips)": 350,
    "8 workgroup barriers": 8 * 78,
}
SHADER_GHZ = 2.4  # MI355X peak engine clock (MI355X_MICROARCH.md); the timeline tool measured ~2.3 GHz sustained


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, choices=[1, 2, 4], default=1, help="BASELINE.json configs[i]: 1 = headline (weak scaling, 256 utterances per GPU); 2, 4 = strong scaling of the named total batch")
    ap.add_argument("--batch", type=int, default=0, help="config 1: utterances per GPU (default 256); configs 2 / 4: TOTAL utterances (default 2048 / 1024)")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--vocab", type=int, default=29)
    ap.add_argument("--beam", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0, help="threads per workgroup (0 = library default)")
    ap.add_argument("--gather", choices=["overlap", "sync", "none"], default="overlap",
                    help="N>1: gather the results to rank 0 inside the timed region; 'overlap' lets batch i's gather "
                         "run (RCCL streams) while batch i+1 is decoded, 'sync' finishes it before the next decode")
    ap.add_argument("--gather-format", choices=["compact", "full"], default="compact",
                    help="what travels to rank 0: 'compact' = valid prefixes only, narrow integer types (expanded on rank 0); "
                         "'full' = the four padded tensors")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and run the gather path even with one rank (self-test)")
    ap.add_argument("--backend", default="", help="torch.distributed backend (default nccl = RCCL; gloo when ranks share a device)")
    ap.add_argument("--min-shard", type=int, default=0, help="configs 2 / 4 (strong scaling): at least this many utterances per rank -- fill a GPU (256 = its CU count) "
                    "before adding ranks; the ranks left over get empty shards (ctcdecode_amd.distributed.shard_size)")
    ap.add_argument("--pmc", action="store_true", default=True, help="(default since round 5) measure roofline.traffic in this run: two rocprofv3 --pmc child runs of "
                    "tools/pmc_child.py (FETCH_SIZE, WRITE_SIZE: counters only, one per pass, as MI355X_MICROARCH.md prescribes; ~8 s each, outside the "
                    "timed region); the constant in --traffic-json is the fallback when the profiler is missing or fails")
    ap.add_argument("--no-pmc", dest="pmc", action="store_false", help="use the committed constant (--traffic-json) for roofline.traffic")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the e2e and other_configs measurements (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="per-launch HBM bytes of the decode kernel from a rocprofv3 --pmc pass (tools/rocprof_summary.py)")
    return ap.parse_args()


def measure_traffic(a):
    """HBM bytes per launch of the headline decode kernel from rocprofv3's FETCH_SIZE / WRITE_SIZE (KiB per dispatch; separate
    passes; the decode kernels read 4-16 B per lane, so FETCH_SIZE needs no 128-bit-load correction: DESIGN.md section 6), each
    pass a child run of this script with the extras off.  Returns (bytes, note) or (None, why)."""
    import csv
    import glob
    import shutil
    import tempfile

    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this run is itself being profiled: no nested counter passes"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    total = 0.0
    tmp = tempfile.mkdtemp(prefix="ctcd_pmc_")
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, c)
            # (the child -- tools/pmc_child.py -- builds the batch this run times: rank 0's, same generator expression, same seed and shape,
            #  decodes it four times and does nothing else; config 1 only: no LM, no pruning)
            cmd = [exe, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable,
                   os.path.join(ROOT, "tools", "pmc_child.py"), str(a.batch or 256), str(a.frames or 1000), str(a.vocab), str(a.beam or 100), str(a.threads), "1234"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=150, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            vals = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "ctc_beam_decode_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, "rocprofv3 --pmc %s gave no rows for the decode kernel (rc %d)" % (c, r.returncode)
            total += sum(vals) / len(vals) * 1024.0
        return int(total), "measured during this run on the batch it times: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/pmc_child.py, which decodes rank 0's batch (same generator, seed 1234, shape) four times; mean over those launches"
    except Exception as e:  # (a profiler that is missing or hangs must not take the bench line with it)
        return None, "rocprofv3 child run failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(lp_np, beam, target_s, lm=None):
    """The reference CPU path (oracle/_ref = its own sources; falls back to this repo's restatement = "port") timed on
    this box's host cores with num_processes = os.cpu_count(), on a bounded sample of the same workload.
    lm = (labels, arpa path, alpha, beta): with the reference's scorer (configs[4])."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_util as ou

    cores = os.cpu_count() or 1
    which, kind = ("reference", "reference") if ou.have_reference() else ("restated", "port")
    kw = dict(beam=beam, cutoff_top_n=lp_np.shape[2], which=which, threads=cores)
    if lm:
        kw["scorer"] = ou.Scorer(lm[2], lm[3], lm[1], lm[0], which)
    n0 = min(lp_np.shape[0], max(cores, 4))
    t0 = time.perf_counter()
    ou.decode(lp_np[:n0], **kw)
    dt = time.perf_counter() - t0
    n, best = n0, n0 / dt
    n1 = min(lp_np.shape[0], int(best * target_s) // cores * cores)
    if n1 > n0 and dt < target_s * 0.6:
        t0 = time.perf_counter()
        ou.decode(lp_np[:n1], **kw)
        dt = time.perf_counter() - t0
        n, best = n1, n1 / dt
    return {"value": round(best, 3), "unit": "utterances/s", "cores": cores, "kind": kind,
            "sample": "%d of the %d utterances of one batch (same T, V, beam), num_processes=%d, %.1f s wall" % (n, lp_np.shape[0], cores, dt)}


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks here (one process per GPU,
    the same environment torch.distributed.run would set), pass rank 0's JSON line through, return the worst exit code."""
    import torch

    ndev = torch.cuda.device_count()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   CTCD_BENCH_VISIBLE_DEVICES=str(ndev))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def synth_rows(torch, B, T, V, seed, kind="randn"):
    """Synthetic log-probability rows (CPU tensor): log_softmax of N(0,1) logits; "blank": +6 on label 0; "peaky": one label
    +8 per frame, the same label for runs of 5-15 frames."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    lg = torch.randn((B, T, V), generator=g)
    if kind == "blank":
        lg[:, :, 0] += 6.0
    elif kind == "peaky":
        runs = torch.randint(5, 16, (B, T // 5 + 1), generator=g)
        labs = torch.randint(0, V, (B, T // 5 + 1), generator=g)
        for b in range(B):
            idx = torch.repeat_interleave(labs[b], runs[b])[:T]
            lg[b, torch.arange(T), idx] += 8.0
    return lg.log_softmax(-1)


def time_e2e(torch, dec, lp_cpu, reps=5, warm=2):
    """SURVEY 8(d)'s primary definition: wall time of one drop-in decode() call -- CPU tensor in, four CPU tensors out."""
    for _ in range(warm):
        dec.decode(lp_cpu)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        dec.decode(lp_cpu)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def time_compact(torch, ctcdecode_amd, dec, lp, steps=10, warm=3):
    """The same launches handing their results over in COMPACT form (what decode() ships over PCIe and what a rank ships to
    the gathering rank): no padded [B, K, T] tensors exist on the device, so there is nothing to zero-fill."""
    B = lp.shape[0]
    tickets = [dec.decode_compact_async(lp, None) for _ in range(warm)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tickets = [dec.decode_compact_async(lp, None) for _ in range(steps)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    dec.finish_compact(tickets[-1])
    return dt


def time_pipelined(torch, ctcdecode_amd, dev, lp, labels, V, K, inflight=2, steps=24, warm=6):
    """The same batches with `inflight` launches in flight (one decoder + stream each): the kernel time of a launch is set by
    its slowest utterance (ties at the beam boundary cost an exact std::nth_element replay), so a lone launch leaves CUs
    idle towards its end; the next launch's workgroups fill them.  An extra, not the headline: `value` times launches
    one at a time, which is what its roofline duration and the rocprofv3 summary describe."""
    decs = [ctcdecode_amd.CTCBeamDecoder(labels, cutoff_top_n=V, beam_width=K, log_probs_input=True, device=dev) for _ in range(inflight)]
    for d in decs:
        d.set_cu_sharing(1)  # the build of the kernel that fits two utterances on a CU: the launches really overlap
    streams = [torch.cuda.Stream(device=dev) for _ in range(inflight)]

    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % inflight]):
                decs[i % inflight].decode_device(lp, None, check=False)
        torch.cuda.synchronize()

    run(warm)
    t0 = time.perf_counter()
    run(steps)
    dt = (time.perf_counter() - t0) / steps
    for d in decs:
        ctcdecode_amd._native.check(ctcdecode_amd._native.lib.ctcd_check_status(d._handle, lp.shape[0]))
    return dt


def time_inflight(torch, ctcdecode_amd, dev, lp, labels, K, inflight, steps=12, warm=4, **dec_kw):
    """Batches of lp.shape[0] utterances through ctcdecode_amd.DecodePipeline with `inflight` launches in flight, DEFAULT build of
    the kernel (one workgroup per CU, nothing shared): seconds per batch.  inflight = 1 is the plain loop."""
    V = lp.shape[2]
    kw = dict(cutoff_top_n=V)
    kw.update(dec_kw)
    pipe = ctcdecode_amd.DecodePipeline(lambda: ctcdecode_amd.CTCBeamDecoder(labels, beam_width=K, log_probs_input=True, device=dev, **kw), inflight=inflight)

    def run(n):  # (results are collected -- and dropped -- as a serving loop would: `inflight` batches are alive at any time)
        tickets = []
        for _ in range(n):
            tickets.append(pipe.submit(lp))
            if len(tickets) > inflight:
                pipe.result(tickets.pop(0))
        for t in tickets:
            pipe.result(t)
        torch.cuda.synchronize()

    run(warm)
    t0 = time.perf_counter()
    run(steps)
    dt = (time.perf_counter() - t0) / steps
    del pipe
    return dt


def arpa_unigrams(path):
    """The words an ARPA file lists (what Scorer::fill_dictionary reads from the model, scorer.cpp:196-230)."""
    words, on = [], False
    with open(path, encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line.startswith("\\"):
                on = line == "\\1-grams:"
                continue
            if on and line:
                words.append(line.split("\t")[1] if "\t" in line else line.split()[1])
    return words


def synth_transcript_rows(torch, B, T, labels, words, seed, boost=7.0):
    """Rows that look like the posteriors of an acoustic model reading sentences: every utterance spells a random sequence of
    the vocabulary's words (each label held for 2-4 frames, blanks between repeated letters and now and then elsewhere, a space
    between words) with +`boost` on the intended label over N(0,1) logits.  The beam then follows word sequences, as it does on
    speech -- N(0,1) rows make it wander through every word history there is."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    idx = {c: i for i, c in enumerate(labels)}
    usable = [w for w in words if w and all(c in idx for c in w) and " " not in w]
    lg = torch.randn((B, T, len(labels)), generator=g)
    for b in range(B):
        t, prev = 0, None
        while t < T:
            w = usable[int(torch.randint(0, len(usable), (1,), generator=g))] + " "
            for c in w:
                if c == prev or int(torch.randint(0, 4, (1,), generator=g)) == 0:  # a blank between equal letters, sometimes elsewhere
                    n = int(torch.randint(1, 3, (1,), generator=g))
                    lg[b, t:t + n, 0] += boost
                    t += n
                n = int(torch.randint(2, 5, (1,), generator=g))
                lg[b, t:t + n, idx[c]] += boost
                t += n
                prev = c
                if t >= T:
                    break
    return lg.log_softmax(-1)


def time_scorer_hook(torch, ctcdecode_amd, dev, arpa, labels, B=128, T=1500, K=100, alpha=0.5, beta=1.0, transcripts=False, threads=1):
    """what the host-side scorer hook costs at the configs[4] per-GPU shape.  The built-in tables of `arpa`
    sit behind the hook as a NATIVE callback (ctcd_scorer_cond_log10 has the callback's signature: no Python in the loop), so
    results and cache contents are the built-in scorer's and the difference in time is the hook's: cold (fresh scorer: every
    window is a miss), a second batch of OTHER utterances (lukewarm), the first batch again (warm: one launch), against the
    same decodes with the built-in tables."""
    import ctypes

    n = ctcdecode_amd._native
    V = len(labels)
    if transcripts:
        voc = [w for w in arpa_unigrams(arpa) if w not in ("<s>", "</s>", "<unk>")]
        lps = [synth_transcript_rows(torch, B, T, labels, voc, 7 + i).to(dev) for i in range(2)]
    else:
        lps = [synth_rows(torch, B, T, V, 7 + i).to(dev) for i in range(2)]
    arr = (ctypes.c_char_p * V)(*[x.encode("utf-8") for x in labels])
    inner = ctypes.c_void_p()
    n.check(n.lib.ctcd_scorer_create(ctypes.byref(inner), 0.0, 0.0, arpa.encode(), arr, V, dev.index or 0))
    order = int(n.lib.ctcd_scorer_max_order(inner))
    fn_addr = ctypes.cast(n.lib.ctcd_scorer_cond_log10, ctypes.c_void_p).value
    out = {"rows": "transcript-like (sentences of the model's words spelled with +7 on the intended label)" if transcripts else "log-softmax of N(0,1) logits (the beam wanders through every word history: the cache's worst case)"}
    try:
        ref = ctcdecode_amd.CTCBeamDecoder(labels, model_path=arpa, alpha=alpha, beta=beta, cutoff_top_n=V, beam_width=K, log_probs_input=True, device=dev)
        ref.set_timing(True)
        want = []
        for lp in lps:
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r = ref.decode_device(lp, None, check=True)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            want.append(r)
        out["built-in tables"] = {"call_ms": round(dt * 1e3, 3), "kernel_ms": round(ref.last_kernel_ms(), 3)}
        sc = ctcdecode_amd.CallbackScorer.from_c(fn_addr, inner.value, arpa_unigrams(arpa), order, labels, alpha=alpha, beta=beta, device=dev)
        if threads > 1:  # (the built-in tables are read-only: the callback may be asked from several threads at once)
            sc.set_callback_threads(threads)
            out["callback_threads"] = threads
        dec = ctcdecode_amd.CTCBeamDecoder(labels, scorer=sc, cutoff_top_n=V, beam_width=K, log_probs_input=True, device=dev)
        calls0, secs0 = 0, 0.0
        for name, lp, w in (("cold (fresh scorer)", lps[0], want[0]), ("second batch of other utterances (lukewarm)", lps[1], want[1]), ("first batch again (warm)", lps[0], want[0])):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = dec.decode_device(lp, None, check=True)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            calls = sc.callback_calls()
            secs = sc.callback_seconds()
            same = all(torch.equal(a, b) for a, b in zip(r, w))
            out[name] = {"call_ms": round(dt * 1e3, 3), "launches": int(n.lib.ctcd_last_scorer_rounds(dec._handle)),
                         "answer_batches_to_waiting_launches": int(n.lib.ctcd_last_scorer_waits(dec._handle)), "callback_calls": int(calls - calls0),
                         "ms_inside_the_callback": round((secs - secs0) * 1e3, 1), "equals_built_in": bool(same)}
            calls0, secs0 = calls, secs
        del dec, sc, ref
    finally:
        n.lib.ctcd_scorer_destroy(inner)
    return out


def time_streaming(torch, ctcdecode_amd, dev, lp, V, K, chunk=50, reps=3):
    """SURVEY 8(f) N3 as a serving loop would drive it: every utterance of the batch is a stream (OnlineCTCBeamDecoder /
    DecoderState), fed in chunks of `chunk` frames from HBM; the last chunk ends the streams and returns the results (CPU
    tensors, as the reference's contract has it).  Reported: the steady-state chunk call (median of the calls in which no
    stream ends: wall clock and the kernel's own time by HIP events), per frame against the one-shot kernel, and the final
    call with its result delivery."""
    import statistics

    B, T, _ = lp.shape
    dec = ctcdecode_amd.OnlineCTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=V, beam_width=K, log_probs_input=True, device=dev)
    ctcdecode_amd._native.check(ctcdecode_amd._native.lib.ctcd_set_timing(dec._handle, 1))
    import ctypes
    best = None
    for _ in range(reps):
        states = [ctcdecode_amd.DecoderState(dec) for _ in range(B)]
        chunks = [lp[:, f0:f0 + chunk].contiguous() for f0 in range(0, T, chunk)]
        torch.cuda.synchronize()
        walls, kerns = [], []
        t_all = time.perf_counter()
        for i, c in enumerate(chunks):
            t0 = time.perf_counter()
            res = dec.decode(c, states, [i == len(chunks) - 1] * B)
            walls.append(time.perf_counter() - t0)
            ms = ctypes.c_float()
            ctcdecode_amd._native.lib.ctcd_last_kernel_ms(dec._handle, ctypes.byref(ms))
            kerns.append(ms.value)
        total = time.perf_counter() - t_all
        del states
        if best is None or total < best[0]:
            best = (total, walls, kerns, res)
    total, walls, kerns, res = best
    # ... and as a loop that does not wait for a chunk's status before it feeds the next (check=False: nothing is read back
    # until a stream ends): the chunks' kernels run back to back on the stream; time of the calls in which no stream ends
    back_to_back = None
    if len(chunks) > 2:
        ctcdecode_amd._native.check(ctcdecode_amd._native.lib.ctcd_set_timing(dec._handle, 0))
        for _ in range(reps):
            states = [ctcdecode_amd.DecoderState(dec) for _ in range(B)]
            dec.decode(chunks[0], states, [False] * B)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for c in chunks[1:-1]:
                dec.decode(c, states, [False] * B, check=False)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (len(chunks) - 2)
            res2 = dec.decode(chunks[-1], states, [True] * B)
            assert int(res2[3].sum()) == int(res[3].sum())
            del states
            back_to_back = dt if back_to_back is None else min(back_to_back, dt)
    # ... and two such groups of streams, each behind its own decoder on its own HIP stream, fed alternately: a chunk's launch lasts as
    # long as its slowest stream (the streams whose chunk holds the most tie frames), and what one group's stragglers leave
    # idle the other group's chunk takes.  Time per chunk of ONE group (= the pair's time / 2).
    two_groups = None
    if len(chunks) > 2:
        try:
            dec2 = ctcdecode_amd.OnlineCTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=V, beam_width=K, log_probs_input=True, device=dev)
            s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            lp2 = torch.roll(lp, 1, 0)  # (the other group decodes the same utterances in another order)
            chunks2 = [lp2[:, f0:f0 + chunk].contiguous() for f0 in range(0, T, chunk)]
            for _ in range(reps):
                st_a = [ctcdecode_amd.DecoderState(dec) for _ in range(B)]
                st_b = [ctcdecode_amd.DecoderState(dec2) for _ in range(B)]
                with torch.cuda.stream(s0):
                    dec.decode(chunks[0], st_a, [False] * B)
                with torch.cuda.stream(s1):
                    dec2.decode(chunks2[0], st_b, [False] * B)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for ca, cb in zip(chunks[1:-1], chunks2[1:-1]):
                    with torch.cuda.stream(s0):
                        dec.decode(ca, st_a, [False] * B, check=False)
                    with torch.cuda.stream(s1):
                        dec2.decode(cb, st_b, [False] * B, check=False)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / (len(chunks) - 2) / 2
                with torch.cuda.stream(s0):
                    ra = dec.decode(chunks[-1], st_a, [True] * B)
                with torch.cuda.stream(s1):
                    rb = dec2.decode(chunks2[-1], st_b, [True] * B)
                torch.cuda.synchronize()
                assert int(ra[3].sum()) == int(res[3].sum()) == int(rb[3].sum())
                del st_a, st_b
                two_groups = dt if two_groups is None else min(two_groups, dt)
            del dec2, chunks2, lp2
        except Exception as e:  # (reported, not fatal: the entry is an extra)
            two_groups = None
            sys.stderr.write("streaming, two groups in flight: %s\n" % str(e)[:200])
    mid_w = statistics.median(walls[1:-1]) if len(walls) > 2 else walls[0]
    mid_k = statistics.median(kerns[1:-1]) if len(kerns) > 2 else kerns[0]
    return {"what": "%d streams fed in %d-frame chunks through OnlineCTCBeamDecoder.decode (HBM-resident input; the last call ends the streams and returns the CPU result tensors)" % (B, chunk),
            "chunk_frames": chunk, "calls": len(walls), "ms_per_chunk_call": round(mid_w * 1e3, 3), "us_per_frame": round(mid_w / chunk * 1e6, 3),
            "kernel_ms_per_chunk": round(mid_k, 3), "kernel_us_per_frame": round(mid_k / chunk * 1e3, 3),
            "back_to_back_ms_per_chunk": round(back_to_back * 1e3, 3) if back_to_back else None,
            "back_to_back_us_per_frame": round(back_to_back / chunk * 1e6, 3) if back_to_back else None,
            "two_groups_in_flight_ms_per_chunk": round(two_groups * 1e3, 3) if two_groups else None,
            "two_groups_in_flight_us_per_frame": round(two_groups / chunk * 1e6, 3) if two_groups else None,
            "final_call_ms": round(walls[-1] * 1e3, 3), "final_call_kernel_ms": round(kerns[-1], 3), "ms_per_batch": round(total * 1e3, 3), "value": round(B / total, 1), "unit": "utterances/s",
            "result_lens_sum": int(res[3].sum())}


def other_configs(torch, ctcdecode_amd, dev, traffic_consts=None):
    """Kernel time of the other BASELINE.json configurations' per-GPU shapes (not bench lines: one or two launches each)."""
    out = {}
    traffic_consts = traffic_consts or {}

    def run(name, B, T, V, K, top_n=40, cutoff_prob=1.0, reps=2, kind="randn", seed=7, prob_input=False, logits=False, **kw):
        lp = synth_rows(torch, B, T, V, seed, kind)
        if prob_input:  # the reference's DEFAULT input mode (log_probs_input=False): probabilities, converted on the device
            lp = lp.exp()
        lp = lp.to(dev)
        if logits:  # logits_input=True (extension): the rows are taken as raw logits (a log-softmax row is a logit row of the same distribution)
            kw = dict(kw, logits_input=True)
        labels = [str(i) for i in range(V)]
        if V == 29:  # blank, apostrophe, space, a..z: the words of tests/data/test.arpa can be spelled
            labels = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
        dec = ctcdecode_amd.CTCBeamDecoder(labels, cutoff_top_n=top_n, cutoff_prob=cutoff_prob, beam_width=K,
                                           log_probs_input=not prob_input, device=dev, **kw)
        dec.set_timing(True)
        ks, ps, ws = [], [], []
        for _ in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = dec.decode_device(lp, None, check=True)
            torch.cuda.synchronize()
            ws.append(time.perf_counter() - t0)
            res_len = res[3].sum().item()
            del res
            ks.append(dec.last_kernel_ms())
            if top_n < V or cutoff_prob < 1.0:
                ps.append(dec.last_prune_ms())
        r = {"B": B, "T": T, "V": V, "beam": K, "decode_kernel_ms": round(min(ks[1:]), 3), "us_per_frame": round(min(ks[1:]) * 1e3 / T, 3),
             "utt_per_s_kernel": round(B / (min(ks[1:]) + (min(ps[1:]) if ps else 0.0)) * 1e3, 1),
             "call_ms": round(min(ws[1:]) * 1e3, 3)}  # the whole HBM-to-HBM call, wall clock (prune pass, tie replay, read-backs included)
        if ps:
            r["prune_kernel_ms"] = round(min(ps[1:]), 3)
            r["prune_GBps"] = round(B * T * V * 4 / (min(ps[1:]) * 1e-3) / 1e9, 1)
            pt = traffic_consts.get("prune_hbm_bytes_per_launch")
            r["prune_roofline"] = {"bound": "hbm", "achieved": r["prune_GBps"], "peak": 8000.0, "unit": "GB/s", "frac": round(r["prune_GBps"] / 8000.0, 4),
                                   "algorithmic_bytes_per_launch": B * T * V * 4, "traffic": pt, "traffic_ratio": round(pt / (B * T * V * 4), 2) if pt else None,
                                   "traffic_source": "profiles/traffic_latest.json (rocprofv3 --pmc, FETCH_SIZE x2 for 128-bit loads: calibrated)"}
            if logits:
                r["prune_kernel_is"] = "prune_logits_wg_kernel: the logits are read once, normalised in registers (sum of exponentials in ctcd_log_softmax's defined order) and only the kept candidates are written -- log-softmax pre-pass + prune in one kernel"
                r["prune_roofline"]["traffic"] = None  # (the PMC constants in profiles/traffic_latest.json are the two-pass prune's)
                r["prune_roofline"]["traffic_ratio"] = None
                # the two-pass form it replaces (debug switch), and the stand-alone normalisation (ctcd_log_softmax: read + write)
                dec.set_fused_logits(False)
                p2 = []
                for _ in range(3):
                    dec.decode_device(lp, None, check=True)
                    p2.append(dec.last_prune_ms())
                dec.set_fused_logits(True)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                outb = torch.empty_like(lp)
                lt = []
                for _ in range(4):
                    ev[0].record()
                    ctcdecode_amd._native.check(ctcdecode_amd._native.lib.ctcd_log_softmax(dec._handle, lp.data_ptr(), None, B, T, V, outb.data_ptr(), torch.cuda.current_stream().cuda_stream))
                    ev[1].record()
                    torch.cuda.synchronize()
                    lt.append(ev[0].elapsed_time(ev[1]))
                del outb
                gb = 2 * B * T * V * 4 / (min(lt[1:]) * 1e-3) / 1e9
                r["two_pass_form"] = {"log_softmax_kernel_ms": round(min(lt[1:]), 3), "separate_prune_kernel_ms": round(min(p2[1:]), 3),
                                      "log_softmax_roofline": {"bound": "hbm", "achieved": round(gb, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gb / 8000.0, 4),
                                                               "algorithmic_bytes_per_launch": 2 * B * T * V * 4}}
            r["prune_flagged_rows"] = int(ctcdecode_amd._native.lib.ctcd_last_prune_flagged_rows(dec._handle))  # settled by the device's std::sort replay ...
            r["prune_host_rows"] = int(ctcdecode_amd._native.lib.ctcd_last_prune_host_rows(dec._handle))  # ... except these
        if K == 500 and not ps:  # the wide-beam kernel's own roofline block
            alg = B * (T * V * 4 + 8 * K + 4) + 8 * int(res_len)
            wt = traffic_consts.get("wide_beam_hbm_bytes_per_launch")
            gbps = alg / (min(ks[1:]) * 1e-3) / 1e9
            r["roofline"] = {"bound": "hbm", "achieved": round(gbps, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(gbps / 8000.0, 5), "algorithmic_bytes_per_launch": alg,
                             "traffic": wt, "traffic_ratio": round(wt / alg, 2) if wt else None, "traffic_source": "profiles/traffic_latest.json (rocprofv3 --pmc passes of tools/bench_configs.py --only 2)"}
        out[name] = r
        del dec, lp
        torch.cuda.empty_cache()

    # N(0,1) logits emit a label on 88 % of the frames -- unrepresentative of real CTC posteriors; the same
    # shape with a blank-dominated distribution (+6 on the blank logit) and a peaky one (one label +8 per frame, in runs of
    # 5-15 frames: the chain-shaped beams of DESIGN.md section 10), and with probabilities as input (the reference's default)
    run("configs[1] shape, blank-dominated rows (+6 on the blank logit)", 256, 1000, 29, 100, top_n=29, kind="blank")
    run("configs[1] shape, peaky rows (one label +8 per frame, runs of 5-15 frames)", 256, 1000, 29, 100, top_n=29, kind="peaky")
    run("configs[1] shape, probability input (log_probs_input=False: device prob -> log pass + decode)", 256, 1000, 29, 100, top_n=29, prob_input=True)
    run("configs[1] shape with 512 utterances per launch (two workgroups per CU)", 512, 1000, 29, 100)
    run("configs[2] per-GPU shape (256 of 2048 utterances, beam 500, T 2000)", 256, 2000, 29, 500, reps=1)
    run("configs[3] (V=10000, top_n 40, cutoff_prob 0.99)", 64, 500, 10000, 100, top_n=40, cutoff_prob=0.99)
    run("configs[3] fed raw logits (logits_input=True: pre-pass + prune fused)", 64, 500, 10000, 100, top_n=40, cutoff_prob=0.99, logits=True)
    run("configs[4] per-GPU shape without the LM (128 of 1024 utterances, T 1500)", 128, 1500, 29, 100)
    # configs[3]'s 64 utterances occupy a quarter of the CUs -- the rest goes to the next batches (DecodePipeline; the rule
    # inflight_for(64) says 2, four launches fill the chip)
    try:
        lp64 = synth_rows(torch, 64, 500, 10000, 7).to(dev)
        lab10k = [str(i) for i in range(10000)]
        q = {"what": "batches of 64 utterances (configs[3]: V=10000, top_n 40, cutoff_prob 0.99; a quarter of the CUs) through ctcdecode_amd.DecodePipeline, prune pass + decode per batch; utterances/s",
             "inflight_for(64)": int(ctcdecode_amd.DecodePipeline.inflight_for(64, dev))}
        for k in (1, 2, 4):
            dt = time_inflight(torch, ctcdecode_amd, dev, lp64, lab10k, 100, k, steps=16, warm=4, cutoff_top_n=40, cutoff_prob=0.99)
            q["%d_in_flight" % k] = {"utt_per_s": round(64 / dt, 1), "ms_per_batch": round(dt * 1e3, 3)}
        out["configs[3], idle three quarters of the chip given to the next batches"] = q
        del lp64
        torch.cuda.empty_cache()
    except Exception as e:
        out["configs[3], several launches in flight"] = {"error": str(e)[:200]}
    arpa = os.path.join(ROOT, "tests", "data", "test.arpa")
    # 128 utterances occupy half the CUs.  The time axis cannot be split, so the idle half is given to the
    # NEXT batch: ctcdecode_amd.DecodePipeline, two launches in flight on two streams, default build, no CU shared.
    try:
        lab29 = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
        lp128 = synth_rows(torch, 128, 1500, 29, 7).to(dev)
        half = {"what": "batches of 128 utterances (configs[4]'s per-GPU share: half the CUs) through ctcdecode_amd.DecodePipeline: one launch at a time vs two in flight on two HIP streams (default build of the kernel, no CU shared); utterances/s"}
        for name, kw in (("no LM", {}), ("LM scorer (tests/data/test.arpa)", dict(model_path=arpa, alpha=0.5, beta=1.0))):
            if kw and not (os.path.exists(arpa) and getattr(ctcdecode_amd, "HAVE_LM", False)):
                continue
            one = time_inflight(torch, ctcdecode_amd, dev, lp128, lab29, 100, 1, **kw)
            two = time_inflight(torch, ctcdecode_amd, dev, lp128, lab29, 100, 2, **kw)
            half[name] = {"one_in_flight": round(128 / one, 1), "two_in_flight": round(128 / two, 1), "ms_per_batch_one": round(one * 1e3, 3), "ms_per_batch_two": round(two * 1e3, 3)}
        out["configs[4] per-GPU shape, idle half of the chip given to the next batch"] = half
        del lp128
    except Exception as e:
        out["configs[4] per-GPU shape, two launches in flight"] = {"error": str(e)[:200]}
    if os.path.exists(arpa) and getattr(ctcdecode_amd, "HAVE_LM", False):
        try:
            run("configs[4] per-GPU shape with the LM scorer (tests/data/test.arpa, alpha 0.5, beta 1.0)", 128, 1500, 29, 100,
                model_path=arpa, alpha=0.5, beta=1.0)
        except Exception as e:  # the LM tier must not take the bench line down
            out["configs[4] with the LM scorer"] = {"error": str(e)[:200]}
    if os.path.exists(arpa) and getattr(ctcdecode_amd, "HAVE_LM", False):
        try:
            lab29 = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
            out["scorer hook at the configs[4] per-GPU shape (tests/data/test.arpa behind a native callback)"] = time_scorer_hook(torch, ctcdecode_amd, dev, arpa, lab29)
            out["scorer hook, transcript-like rows (tests/data/test.arpa behind a native callback)"] = time_scorer_hook(torch, ctcdecode_amd, dev, arpa, lab29, transcripts=True)
        except Exception as e:
            out["scorer hook (test.arpa)"] = {"error": str(e)[:300]}
    # the same shape with a language model of realistic size (generated: 50 000 words, 3-gram, ~20 MB of ARPA text; the
    # tables are tens of MB, so the kernel's look-ups miss L2 -- test.arpa's 37 unigrams never do)
    if getattr(ctcdecode_amd, "HAVE_LM", False):
        try:
            import importlib.util
            import tempfile

            spec = importlib.util.spec_from_file_location("make_big_lm", os.path.join(ROOT, "tools", "make_big_lm.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            big = os.path.join(tempfile.gettempdir(), "ctcd_big_words_50k.arpa")
            if not os.path.exists(big):
                mod.make(big)
            run("configs[4] per-GPU shape with a generated 50k-word 3-gram LM (alpha 0.5, beta 1.0)", 128, 1500, 29, 100, model_path=big, alpha=0.5, beta=1.0, reps=1)
            try:
                out["scorer hook, transcript-like rows (generated 50k-word model behind a native callback)"] = time_scorer_hook(
                    torch, ctcdecode_amd, dev, big, ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)], transcripts=True)
                out["scorer hook, the same with eight callback threads (ctcd_scorer_set_callback_threads: the built-in tables are read-only)"] = time_scorer_hook(
                    torch, ctcdecode_amd, dev, big, ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)], transcripts=True, threads=8)
            except Exception as e:
                out["scorer hook (50k-word model)"] = {"error": str(e)[:300]}
        except Exception as e:
            out["configs[4] with the generated 50k-word LM"] = {"error": str(e)[:200]}
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a))
    import numpy as np
    import torch

    import ctcdecode_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (a.gpus, world, a.gpus))
    ndev = torch.cuda.device_count()
    if ndev == 0:  # (the hot path has no CPU form: fail loudly instead of measuring something else)
        raise SystemExit("bench.py: no HIP device visible -- this benchmark times the MI355X kernels and has no CPU fallback")
    shared = ndev < world  # fewer devices than ranks (single-GPU dry run of the N-rank path): ranks share devices
    torch.cuda.set_device(local % ndev)
    dev = torch.device("cuda", local % ndev)
    dist = None
    use_dist = world > 1 or a.force_dist
    # RCCL prints a version banner on stdout; keep stdout for the ONE JSON line: everything else goes to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    backend = a.backend or ("gloo" if shared else "nccl")  # RCCL refuses two ranks on one device
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            # (no device_id=: binding the group to the device at init makes every collective wait for ALL work queued on
            #  the device -- the next batch's decode kernel included -- instead of for its own stream only)
            dist.init_process_group("nccl", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # BASELINE.json configs[a.config]: (total or per-GPU batch, frames, beam, scaling, LM)
    cfg = {1: dict(batch=256, frames=1000, beam=100, scaling="weak", lm=None),
           2: dict(batch=2048, frames=2000, beam=500, scaling="strong", lm=None),
           4: dict(batch=1024, frames=1500, beam=100, scaling="strong", lm=(os.path.join(ROOT, "tests", "data", "test.arpa"), 0.5, 1.0))}[a.config]
    T, V, K = a.frames or cfg["frames"], a.vocab, a.beam or cfg["beam"]
    strong = cfg["scaling"] == "strong"
    B_total = (a.batch or cfg["batch"]) * (1 if strong else world)
    if strong:  # contiguous blocks of ceil(B / world) utterances, as decode_sharded cuts them
        from ctcdecode_amd.distributed import shard_bounds

        lo, hi = shard_bounds(B_total, world, rank, a.min_shard)
        B = hi - lo
    else:
        B = a.batch or cfg["batch"]
    labels = [str(i) for i in range(V)]
    dec_kw = {}
    if cfg["lm"]:  # blank, apostrophe, space, a..z: the words of tests/data/test.arpa can be spelled
        assert V == 29
        labels = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
        dec_kw = dict(model_path=cfg["lm"][0], alpha=cfg["lm"][1], beta=cfg["lm"][2])
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    lp_cpu = torch.randn((max(B, 1), T, V), generator=g, dtype=torch.float32).log_softmax(-1)[:B]
    lp = lp_cpu.to(dev)

    def make_decoder():
        d = ctcdecode_amd.CTCBeamDecoder(labels, cutoff_top_n=V, beam_width=K, blank_id=0, log_probs_input=True, device=dev, **dec_kw)
        if a.threads:
            d.set_threads(a.threads)
        return d

    dec = make_decoder()
    dec.set_timing(True)

    gatherer = None
    compact = use_dist and a.gather != "none" and a.gather_format == "compact"
    # N > 1, device backend, overlapped compact gather: the host looks at batch i (status words, label count, the gather's
    # bookkeeping; on rank 0 the expansion) only AFTER batch i+1's kernel is queued, and the collectives run on a side
    # stream behind batch i's own event -- the kernels still run one after another on one stream, as at N = 1, but the
    # GPU never waits for the host.  Two decoder objects alternate (a decoder's workspace and compact buffers belong to
    # the batch in flight).
    lookahead = compact and a.gather == "overlap" and backend != "gloo"
    if use_dist and a.gather != "none":
        from ctcdecode_amd.distributed import make_gatherer

        from ctcdecode_amd.distributed import shard_size
        per = shard_size(B_total, world, a.min_shard) if strong else B  # the padded shard size of the "full" format
        gatherer = make_gatherer(a.gather_format, per, K, T, V, dev, dst=0, depth=2, decoder=dec,
                                 stream=torch.cuda.Stream(device=dev) if lookahead else None)
    decs = [dec]
    if lookahead:
        decs.append(make_decoder())
    pending = []  # [(decoder, ticket)] of the batch whose kernel is queued but whose results the host has not handled yet
    nlaunch = [0]

    def handle_pending():
        while pending:
            d, ticket = pending.pop(0)
            gatherer.submit(d.finish_compact(ticket), ready=ticket["event"])

    def step():
        # N > 1 with the compact gather: a rank produces its results in compact form (nothing padded is written on it);
        # rank 0 rebuilds the padded tensors of ALL ranks in its HBM.  N = 1: the padded tensors are written by the decode.
        if lookahead:
            gatherer.wait()  # (the gathers of two batches ago read this decoder's buffers: long finished, now confirmed)
            d = decs[nlaunch[0] % 2]
            nlaunch[0] += 1
            ticket = d.decode_compact_async(lp, None)
            handle_pending()
            pending.append((d, ticket))
            return None
        res = dec.decode_compact(lp, None) if compact else dec.decode_device(lp, None, check=False)
        if gatherer is not None:
            gatherer.submit(res)
            if a.gather == "sync":
                gatherer.wait()
        return res

    def fence():
        if gatherer is not None:
            handle_pending()
            gatherer.wait()  # every submitted gather has completed (part of the timed work)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[dev.index]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        res = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # kernel duration of the last timed launch (HIP events on the launch stream); averaged over a few extra launches
    # OUTSIDE the timed region so the event reads do not perturb it
    durs = []
    for _ in range(max(3, min(a.steps, 10)) if a.config == 1 else 2):
        res = dec.decode_device(lp, None, check=False)
        torch.cuda.synchronize()
        durs.append(dec.last_kernel_ms())
    ctcdecode_amd._native.check(ctcdecode_amd._native.lib.ctcd_check_status(dec._handle, B))
    kern_ms = float(np.mean(durs))
    kern_all = [kern_ms]
    if use_dist:  # every rank's kernel duration (rank order)
        kt = torch.tensor([kern_ms], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        parts = [torch.zeros_like(kt) for _ in range(world)]
        dist.all_gather(parts, kt)
        kern_all = [float(p.item()) for p in parts]

    out_len = res[3]
    # ALGORITHMIC bytes per utterance: read T*V*4 of input + write the valid token/timestep prefixes
    # (8 bytes per emitted label per beam) + scores and lengths (8 bytes per beam) + 4 (seq_len)
    alg_bytes = B * (T * V * 4 + 8 * K + 4) + 8 * int(out_len.sum().item())
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    traffic, traffic_src, tj = None, None, {}
    if os.path.exists(a.traffic_json):
        try:
            tj = json.load(open(a.traffic_json))
            traffic, traffic_src = tj.get("hbm_bytes_per_launch"), tj.get("source")
        except Exception:
            traffic = None
    traffic_note = ("constant from profiles/%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of this command), not measured in this run" % traffic_src) if traffic else None
    if a.pmc and rank == 0 and a.config == 1 and not use_dist:
        measured, note = measure_traffic(a)
        if measured:
            traffic, traffic_note = measured, note
        elif traffic_note:
            traffic_note += "; --pmc: " + note

    if rank == 0:
        floor = sum(FRAME_FLOOR.values())
        frame_clocks = kern_ms * 1e-3 / T * SHADER_GHZ * 1e9
        workload = {1: "BASELINE.json configs[1]: CTC prefix beam search, no LM, log-softmax of N(0,1) logits; HBM-resident tensors (input in HBM, the four result tensors left in HBM: `e2e` is the host-to-host call)",
                    2: "BASELINE.json configs[2]: B=2048 batch-sharded, beam 500, T=2000, no LM, compact RCCL gather",
                    4: "BASELINE.json configs[4]: B=1024 batch-sharded, beam 100, T=1500, LM scorer on tests/data/test.arpa (alpha 0.5, beta 1.0)"}[a.config]
        line = {
            "metric": "utterances/sec at B=256 T=1000 V=29 beam=100" if a.config == 1 else
                      "utterances/sec at B=%d T=%d V=%d beam=%d%s (BASELINE.json configs[%d])" % (B_total, T, V, K, " with the LM scorer" if cfg["lm"] else "", a.config),
            "value": round(B_total * a.steps / elapsed, 3),
            "unit": "utterances/s",
            # ranks that took part in the run (torch.distributed's world size once the process group is up)
            "n_gpus": dist.get_world_size() if use_dist else 1, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "utterances_per_gpu": B, "frames": T, "vocab": V, "beam_width": K, "cutoff_top_n": V,
                       "global_batch": B_total, "ranks": world, "devices_visible": ndev,
                       "parallelism": "batch-sharded x%d, gather=%s/%s, backend=%s" % (world, a.gather, a.gather_format, backend) if use_dist else "single GPU",
                       "threads_per_workgroup": a.threads or "default"},
            "kernel_ms": round(kern_ms, 4),
            "kernel_ms_per_rank": [round(v, 4) for v in kern_all],
            "us_per_frame": round(kern_ms * 1e3 / T, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 6), "traffic": traffic if a.config == 1 else None,
                         "traffic_source": traffic_note,
                         "traffic_ratio": round(traffic / alg_bytes, 2) if traffic and a.config == 1 else None,
                         "memset_bytes_outside_kernel": 2 * B * K * T * 4 + 2 * B * K * 4,
                         "kernel": "ctc_beam_decode_kernel", "algorithmic_bytes_per_launch": alg_bytes},
            # the recurrence is latency-bound, not HBM-bound: distance from the critical-path floor of one frame
            "latency": {"frame_clocks": round(frame_clocks), "floor_clocks": floor, "frame_over_floor": round(frame_clocks / floor, 2),
                        "shader_ghz_assumed": SHADER_GHZ, "floor_terms": FRAME_FLOOR},
        }
        if strong and not a.no_extras:
            # The partition rule: an utterance occupies one CU, so a shard below the CU count idles CUs
            # without finishing sooner.  This GPU's kernel time for the shard of the named 8-GPU partition and for a shard
            # that fills it: equal times mean half the GPUs do the same job in the same time (--min-shard 256).
            try:
                probe = {}
                for nb in sorted({max(1, B_total // 8), min(B_total, 256)}):
                    lpp = lp[:nb] if nb <= lp.shape[0] else torch.cat([lp] * ((nb + lp.shape[0] - 1) // lp.shape[0]))[:nb]
                    ks = []
                    for _ in range(3):
                        dec.decode_device(lpp, None, check=False)
                        torch.cuda.synchronize()
                        ks.append(dec.last_kernel_ms())
                    probe["%d utterances on one GPU" % nb] = {"kernel_ms": round(min(ks[1:]), 3), "utterances_per_s": round(nb / (min(ks[1:]) * 1e-3), 1)}
                line["partition_probe"] = {"what": "shard of the named 8-rank partition (B/8) vs a shard that fills the GPU's 256 CUs; rule: --min-shard 256 (shard_size)", **probe}
            except Exception as e:
                line["partition_probe"] = {"error": str(e)[:200]}
        if shared:
            line["config"]["oversubscribed"] = "%d ranks on %d device(s): a dry run of the N-rank path, not a scaling number" % (world, ndev)
        if world == 1 and not a.no_extras and a.config == 1:
            # (its own decoder, released afterwards: decode() keeps a second HIP stream for the streamed input, and HIP maps
            #  streams onto a few hardware queues -- 4 by default, GPU_MAX_HW_QUEUES -- so a stream left behind here can end up
            #  sharing a queue with one of the `pipelined` measurement's three: INTEGRATION.md 5)
            dec_e2e = make_decoder()
            e2e = time_e2e(torch, dec_e2e, lp_cpu)
            del dec_e2e
            import gc
            gc.collect()
            line["e2e"] = {"what": "drop-in decode(): CPU float32 tensor (pageable) in, four CPU tensors out (SURVEY 8(d) primary definition); the input streams to the kernel in slices while it decodes and the kernel mirrors its compact results into page-locked host memory (DESIGN.md 2c)",
                           "ms_per_batch": round(e2e * 1e3, 3), "value": round(B / e2e, 1), "unit": "utterances/s"}
            try:
                cp = time_compact(torch, ctcdecode_amd, dec, lp)
                ct = tj.get("compact_hbm_bytes_per_launch")
                line["compact_results"] = {"what": "the same launches with the results left in HBM in compact form (what decode() and the multi-GPU gather ship): no padded tensors on the device, nothing to zero-fill",
                                           "ms_per_batch": round(cp * 1e3, 3), "value": round(B / cp, 1), "unit": "utterances/s", "memset_bytes_outside_kernel": 0,
                                           "traffic": ct, "traffic_ratio": round(ct / alg_bytes, 2) if ct else None}
            except Exception as e:
                line["compact_results"] = {"error": str(e)[:200]}
            try:
                pl = time_pipelined(torch, ctcdecode_amd, dev, lp, [str(i) for i in range(V)], V, K, inflight=3)
                line["pipelined"] = {"what": "the same batches with 3 launches in flight on 3 streams, two-workgroups-per-CU build of the kernel (a serving loop; not the headline: see DESIGN.md 6)",
                                     "launches_in_flight": 3, "ms_per_batch": round(pl * 1e3, 3), "value": round(B / pl, 1), "unit": "utterances/s"}
            except Exception as e:
                line["pipelined"] = {"error": str(e)[:200]}
            try:
                tw = time_inflight(torch, ctcdecode_amd, dev, lp, [str(i) for i in range(V)], K, 2, steps=20, warm=4)
                line["two_in_flight"] = {"what": "the same batches through ctcdecode_amd.DecodePipeline, two launches in flight on two streams, DEFAULT build (one workgroup per CU, nothing shared): a launch lasts as long as its slowest utterance (the ones with the most exact nth_element replays: profiles/r05a_utt_spread.txt), the next batch's workgroups take the CUs the finished ones free.  An extra like `pipelined`: `value` times launches one at a time",
                                         "ms_per_batch": round(tw * 1e3, 3), "value": round(B / tw, 1), "unit": "utterances/s"}
            except Exception as e:
                line["two_in_flight"] = {"error": str(e)[:200]}
            try:  # seeds {0, 1, 2} for the headline shape (kernel time by HIP events, three launches each)
                vals = []
                for sd in (0, 1, 2):
                    lps = synth_rows(torch, B, T, V, sd).to(dev)
                    ks = []
                    for _ in range(4):
                        dec.decode_device(lps, None, check=False)
                        torch.cuda.synchronize()
                        ks.append(dec.last_kernel_ms())
                    vals.append(B / (min(ks[1:]) * 1e-3))
                    del lps
                vals.sort()
                line["seeds"] = {"what": "the headline shape on seeds 0, 1, 2 (kernel time): utterances/s min / median / max", "min": round(vals[0], 1), "median": round(vals[1], 1), "max": round(vals[2], 1)}
            except Exception as e:
                line["seeds"] = {"error": str(e)[:200]}
            try:
                line["streaming"] = time_streaming(torch, ctcdecode_amd, dev, lp, V, K, chunk=50)
                line["streaming"]["one_shot_kernel_us_per_frame"] = round(kern_ms * 1e3 / T, 3)
            except Exception as e:
                line["streaming"] = {"error": str(e)[:200]}
            try:
                line["other_configs"] = other_configs(torch, ctcdecode_amd, dev, tj)
            except Exception as e:
                line["other_configs"] = {"error": str(e)[:300]}
        if not a.no_cpu_baseline:  # (rank 0's host cores, on rank 0's own shard; outside the timed region)
            line["cpu_baseline"] = cpu_baseline(lp_cpu.numpy(), K, a.cpu_seconds, lm=(labels,) + cfg["lm"] if cfg["lm"] else None)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier(device_ids=[dev.index]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
