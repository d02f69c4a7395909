#!/usr/bin/env python3
"""bench.py -- utterances/s of CTC prefix beam-search decoding on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1], per GPU): B=256 utterances, T=1000 frames, V=29 labels, beam_width=100,
cutoff_top_n=29 (no pruning), no LM; inputs are float32 log-softmax of N(0,1) logits, resident in HBM before the
timed region.  A "step" = one decode of that whole batch through the product path (C ABI -> HIP kernel).  With N GPUs
every rank decodes its own B utterances (weak scaling, no data-path collective) and rank 0 then gathers the four result
tensors over RCCL (north_star's "trivial gather") inside the timed region; by default batch i's gather overlaps the
decode of batch i+1 (ctcdecode_amd.distributed.ResultGatherer), all gathers complete before the clock stops.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the definition of every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--vocab", type=int, default=29)
    ap.add_argument("--beam", type=int, default=100)
    ap.add_argument("--threads", type=int, default=0, help="threads per workgroup (0 = library default)")
    ap.add_argument("--gather", choices=["overlap", "sync", "none"], default="overlap",
                    help="N>1: gather the four result tensors to rank 0 inside the timed region; 'overlap' lets batch i's gather "
                         "run (RCCL streams) while batch i+1 is decoded, 'sync' finishes it before the next decode")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and run the gather path even with one rank (self-test)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="per-launch HBM bytes of the decode kernel from a rocprofv3 --pmc pass (tools/rocprof_summary.py)")
    return ap.parse_args()


def cpu_baseline(lp_np, beam, target_s):
    """The reference CPU path (oracle/_ref = its own sources; falls back to this repo's restatement = "port") timed on
    this box's host cores with num_processes = os.cpu_count(), on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_util as ou

    cores = os.cpu_count() or 1
    which, kind = ("reference", "reference") if ou.have_reference() else ("restated", "port")
    n0 = min(lp_np.shape[0], max(cores, 4))
    t0 = time.perf_counter()
    ou.decode(lp_np[:n0], beam=beam, cutoff_top_n=lp_np.shape[2], which=which, threads=cores)
    dt = time.perf_counter() - t0
    n, best = n0, n0 / dt
    n1 = min(lp_np.shape[0], int(best * target_s) // cores * cores)
    if n1 > n0 and dt < target_s * 0.6:
        t0 = time.perf_counter()
        ou.decode(lp_np[:n1], beam=beam, cutoff_top_n=lp_np.shape[2], which=which, threads=cores)
        dt = time.perf_counter() - t0
        n, best = n1, n1 / dt
    return {"value": round(best, 3), "unit": "utterances/s", "cores": cores, "kind": kind,
            "sample": "%d of the %d utterances of one batch (same T, V, beam), num_processes=%d, %.1f s wall" % (n, lp_np.shape[0], cores, dt)}


def main():
    a = parse()
    import numpy as np
    import torch

    import ctcdecode_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (a.gpus, world, a.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    use_dist = world > 1 or a.force_dist
    # RCCL prints a version banner on stdout; keep stdout for the ONE JSON line: everything else goes to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    B, T, V, K = a.batch, a.frames, a.vocab, a.beam
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    lp_cpu = torch.randn((B, T, V), generator=g, dtype=torch.float32).log_softmax(-1)
    lp = lp_cpu.to(dev)
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], cutoff_top_n=V, beam_width=K, blank_id=0, log_probs_input=True, device=dev)
    if a.threads:
        dec.set_threads(a.threads)
    dec.set_timing(True)

    gatherer = None
    if use_dist and a.gather != "none":
        from ctcdecode_amd.distributed import ResultGatherer

        shapes = [((B, K, T), torch.int32), ((B, K), torch.float32), ((B, K, T), torch.int32), ((B, K), torch.int32)]
        gatherer = ResultGatherer(shapes, dev, dst=0, depth=2)

    def step():
        res = dec.decode_device(lp, None, check=False)
        if gatherer is not None:
            gatherer.submit(res)
            if a.gather == "sync":
                gatherer.wait()
        return res

    def fence():
        if gatherer is not None:
            gatherer.wait()  # every submitted gather has completed (part of the timed work)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
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
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # kernel duration of the last timed launch (HIP events on the launch stream); averaged over a few extra launches
    # OUTSIDE the timed region so the event reads do not perturb it
    durs = []
    for _ in range(max(3, min(a.steps, 10))):
        res = dec.decode_device(lp, None, check=False)
        torch.cuda.synchronize()
        durs.append(dec.last_kernel_ms())
    ctcdecode_amd._native.check(ctcdecode_amd._native.lib.ctcd_check_status(dec._handle, B))
    kern_ms = float(np.mean(durs))

    out_len = res[3]
    # ALGORITHMIC bytes per utterance: read T*V*4 of input + write the valid token/timestep prefixes
    # (8 bytes per emitted label per beam) + scores and lengths (8 bytes per beam) + 4 (seq_len)
    alg_bytes = B * (T * V * 4 + 8 * K + 4) + 8 * int(out_len.sum().item())
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic = None
    if os.path.exists(a.traffic_json):
        try:
            traffic = json.load(open(a.traffic_json)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        line = {
            "metric": "utterances/sec at B=256 T=1000 V=29 beam=100",
            "value": round(world * B * a.steps / elapsed, 3),
            "unit": "utterances/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: CTC prefix beam search, no LM, log-softmax of N(0,1) logits",
                       "utterances_per_gpu": B, "frames": T, "vocab": V, "beam_width": K, "cutoff_top_n": V,
                       "global_batch": world * B, "parallelism": "batch-sharded x%d, gather=%s" % (world, a.gather if use_dist else "n/a"),
                       "threads_per_workgroup": a.threads or "default"},
            "kernel_ms": round(kern_ms, 4),
            "us_per_frame": round(kern_ms * 1e3 / T, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 6), "traffic": traffic,
                         "kernel": "ctc_beam_decode_kernel", "algorithmic_bytes_per_launch": alg_bytes},
        }
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(lp_cpu.numpy(), K, a.cpu_seconds)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
