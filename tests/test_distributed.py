"""The N>1 path (ctcdecode_amd/distributed.py) with world_size 2 over gloo on the CPU.  The per-rank decode is stood in
for by the oracle (there is no GPU here); what is tested is the sharding arithmetic and the gather."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import oracle_util as ou

    # the module under test has no GPU dependency of its own; import it without the package __init__ (which needs HIP)
    import importlib.util

    spec = importlib.util.spec_from_file_location("ctcd_distributed", os.path.join(ROOT, "ctcdecode_amd", "distributed.py"))
    dd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dd)

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, V, K = 40, 9, 6
    lp = torch.from_numpy(ou.synth_logprobs(B, T, V, 77))
    sl = torch.tensor([(7 * i) % (T + 3) for i in range(B)], dtype=torch.int32)

    def decode_fn(p, s):
        r = ou.decode(p.numpy(), s.numpy() if s is not None else None, beam=K, threads=1)
        return (torch.from_numpy(r["tokens"]), torch.from_numpy(r["scores"]), torch.from_numpy(r["timesteps"]), torch.from_numpy(r["lens"]))

    got = dd.decode_sharded(decode_fn, lp, sl, dst=0)
    # ... and with the partition rule "fill a GPU before adding ranks" (min_shard: here 4 utterances -- rank 1 gets what is
    # left, possibly nothing)
    got2 = dd.decode_sharded(decode_fn, lp, sl, dst=0, min_shard=4)
    ok = True
    if rank == 0:
        want = ou.decode(lp.numpy(), sl.numpy(), beam=K, threads=1)
        ok = all(np.array_equal(a.numpy(), b.numpy()) for a, b in zip(got, got2))
        ok = ok and (np.array_equal(got[0].numpy(), want["tokens"]) and np.array_equal(got[1].numpy().view(np.uint32), want["scores"].view(np.uint32))
              and np.array_equal(got[2].numpy(), want["timesteps"]) and np.array_equal(got[3].numpy(), want["lens"]))
    else:
        ok = got is None and got2 is None
    lo, hi = dd.shard_bounds(B, world, rank)
    ok = ok and 0 <= lo <= hi <= B
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 4, 1])
def test_sharded_decode_gloo_world2(B):
    import torch.multiprocessing as mp

    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    procs = [mp.get_context("spawn").Process(target=_worker, args=(r, world, port, B, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}


def test_shard_bounds_cover_the_batch():
    import importlib.util

    spec = importlib.util.spec_from_file_location("ctcd_distributed", os.path.join(ROOT, "ctcdecode_amd", "distributed.py"))
    dd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dd)
    for B in [0, 1, 7, 8, 255, 256, 2048]:
        for world in [1, 2, 4, 8]:
            seen = []
            for r in range(world):
                lo, hi = dd.shard_bounds(B, world, r)
                seen += list(range(lo, hi))
            assert seen == list(range(B))
            # the partition rule: shards of at least min_shard fill ranks from the front, the rest stay empty
            seen, sizes = [], []
            for r in range(world):
                lo, hi = dd.shard_bounds(B, world, r, min_shard=256)
                seen += list(range(lo, hi))
                sizes.append(hi - lo)
            assert seen == list(range(B)) and all(sz in (0, B % 256 or 256, 256) or sz == -(-B // world) for sz in sizes)
    assert [dd.shard_bounds(1024, 8, r, min_shard=256) for r in range(8)] == [(0, 256), (256, 512), (512, 768), (768, 1024)] + [(1024, 1024)] * 4


def _gatherer_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import importlib.util

    import torch
    import torch.distributed as dist

    spec = importlib.util.spec_from_file_location("ctcd_distributed", os.path.join(ROOT, "ctcdecode_amd", "distributed.py"))
    dd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dd)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = [((3, 2, 5), torch.int32), ((3, 2), torch.float32)]
    g = dd.ResultGatherer(shapes, torch.device("cpu"), dst=0, depth=2)
    ok = True
    slots = []
    for step in range(5):  # more submissions than receive slots: slots are recycled in order
        res = (torch.full((3, 2, 5), 100 * step + rank, dtype=torch.int32), torch.full((3, 2), float(10 * step + rank)))
        slots.append(g.submit(res))
        if step % 2 == 1:
            g.wait()
            if rank == 0:
                parts = g.received(slots[-1])
                ok = ok and all(int(parts[0][r][0, 0, 0]) == 100 * step + r and float(parts[1][r][0, 0]) == 10 * step + r for r in range(world))
    g.wait()
    if rank == 0:
        parts = g.received(slots[-1])
        ok = ok and all(int(parts[0][r][0, 0, 0]) == 400 + r for r in range(world))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gatherer_gloo_world2():
    import torch.multiprocessing as mp

    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    procs = [mp.get_context("spawn").Process(target=_gatherer_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}


def _gpu_worker(rank, world, port, B, ret):
    """Both ranks on cuda:0 (the GPU box has one device), gloo rendezvous: the per-rank decode is the real HIP path."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import oracle_util as ou

    import ctcdecode_amd
    from ctcdecode_amd import distributed as dd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, V, K = 120, 29, 24
    lp_np = ou.synth_logprobs(B, T, V, 4242)
    sl_np = np.array([(37 * i) % (T + 5) for i in range(B)], np.int32)
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], beam_width=K, log_probs_input=True, device="cuda:0")

    def decode_fn(p, s):  # HIP decode of this rank's block; gloo gathers host tensors
        return tuple(t.cpu() for t in dec.decode_device(p, s))

    got = dd.decode_sharded(decode_fn, torch.from_numpy(lp_np), torch.from_numpy(sl_np), dst=0)
    ok = got is None
    if rank == 0:
        want = ou.decode(lp_np, sl_np, beam=K)
        mine = dict(tokens=got[0].numpy(), scores=got[1].numpy(), timesteps=got[2].numpy(), lens=got[3].numpy(), nres=want["nres"])
        ou.assert_same(mine, want, "decode_sharded over 2 ranks")
        ok = True
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("B", [7, 2])
def test_sharded_decode_real_hip_world2(B):
    import torch.multiprocessing as mp

    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    procs = [mp.get_context("spawn").Process(target=_gpu_worker, args=(r, 2, port, B, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}


def _gpu_compact_worker(rank, world, port, ret):
    """Two ranks on cuda:0 over gloo: each decodes its own batch in compact form, rank 0 gathers and expands both."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import oracle_util as ou

    import ctcdecode_amd
    from ctcdecode_amd import distributed as dd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, T, V, K = 6, 150, 29, 32
    dec = ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], beam_width=K, log_probs_input=True, device="cuda:0")
    g = dd.make_gatherer("compact", B, K, T, V, torch.device("cuda:0"), dst=0, depth=2, decoder=dec)
    ok = True
    for step in range(3):  # more submissions than pipeline slots
        lp = [ou.synth_logprobs(B, T, V, 9000 + 10 * step + r) for r in range(world)]
        g.submit(dec.decode_compact(torch.from_numpy(lp[rank])))
        if step == 2:
            g.wait()
            if rank == 0:
                out, sc, ts, ln = (t.cpu().numpy() for t in g.last)
                want = ou.decode(np.concatenate(lp, 0), beam=K)
                got = dict(tokens=out, scores=sc, timesteps=ts, lens=ln, nres=want["nres"])
                ou.assert_same(got, want, "compact gather over 2 ranks")
                ok = bool(np.array_equal(out, want["tokens"]) and np.array_equal(ts, want["timesteps"]))
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_compact_gather_real_hip_world2():
    import torch.multiprocessing as mp

    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    procs = [mp.get_context("spawn").Process(target=_gpu_compact_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: bench.py starts the two ranks itself (they share the one device
    of this box over gloo -- a dry run of the N-rank path) and rank 0 prints the one JSON line."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "32",
                        "--frames", "200", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["config"]["ranks"] == 2 and d["value"] > 0 and d["config"]["global_batch"] == 64


@pytest.mark.gpu
@pytest.mark.parametrize("config,batch,extra", [(2, 12, ["--beam", "500", "--frames", "120"]), (4, 10, ["--frames", "150"])])
def test_bench_strong_scaling_configs_two_rank_dry_run(config, batch, extra):
    """BASELINE.json's 8-GPU configurations as bench modes (configs[2]: beam 500; configs[4]: with the LM scorer), strong
    scaling: the named TOTAL batch is cut into contiguous blocks (shard_bounds).  Dry run with two ranks on this box's one
    device (gloo), shrunk in batch / frames; cpu_baseline is part of every line."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", str(config), "--steps", "2", "--warmup", "1",
                        "--batch", str(batch), "--cpu-seconds", "2"] + extra, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and d["n_gpus"] == 2 and d["config"]["global_batch"] == batch and d["config"]["utterances_per_gpu"] == batch // 2
    assert d["value"] > 0 and len(d["kernel_ms_per_rank"]) == 2 and d["cpu_baseline"]["value"] > 0
    assert ("configs[%d]" % config) in d["metric"]


@pytest.mark.gpu
def test_compact_gather_one_decoder_depth_two_device_backend():
    """decode_compact returns an owned label tensor, so ONE decoder feeding a depth-2 gatherer on a device backend
    (one-rank RCCL group) cannot have batch i's labels overwritten by batch i+1's decode while they are being gathered."""
    import subprocess

    code = (
        "import os,sys,numpy as np,torch,torch.distributed as dist\n"
        "sys.path.insert(0,%r); sys.path.insert(0,os.path.join(%r,'tests'))\n"
        "import oracle_util as ou, ctcdecode_amd\n"
        "from ctcdecode_amd import distributed as dd\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "B,T,V,K=6,150,29,32\n"
        "dev=torch.device('cuda:0')\n"
        "dec=ctcdecode_amd.CTCBeamDecoder([str(i) for i in range(V)], beam_width=K, log_probs_input=True, device=dev)\n"
        "g=dd.make_gatherer('compact',B,K,T,V,dev,dst=0,depth=2,decoder=dec,stream=torch.cuda.Stream(device=dev))\n"
        "lps=[ou.synth_logprobs(B,T,V,9100+i) for i in range(4)]\n"
        "outs=[]\n"
        "for i,lp in enumerate(lps):\n"
        "    g.submit(dec.decode_compact(torch.from_numpy(lp)))\n"
        "g.wait(); torch.cuda.synchronize()\n"
        "out,sc,ts,ln=(t.cpu().numpy() for t in g.last)\n"
        "want=ou.decode(lps[-1],beam=K)\n"
        "ou.assert_same(dict(tokens=out,scores=sc,timesteps=ts,lens=ln,nres=want['nres']),want,'last batch')\n"
        "dist.destroy_process_group(); print('OK')\n") % (ROOT, ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


@pytest.mark.gpu
def test_bench_gather_behind_next_decode_rccl_one_rank():
    """The N > 1 loop of bench.py on a one-rank RCCL group (the GPU box has one device; RCCL refuses two ranks on it): async
    compact decode tickets, status words fetched without draining the stream, collectives + expansion on a side stream."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--backend", "nccl", "--steps", "4", "--warmup", "2",
                        "--batch", "64", "--frames", "300", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["value"] > 0 and "gather=overlap/compact, backend=nccl" in d["config"]["parallelism"]
