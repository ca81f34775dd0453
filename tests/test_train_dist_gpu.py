"""Data-parallel training step on the GPU with two ranks (both on cuda:0, gloo backend so that one card suffices):
gradients are mean-all-reduced bucket by bucket from the backward sweep, every rank clips and steps identically."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tdvc_amd import synth
    from tdvc_amd.model.pnet import VideoCompressor
    from tdvc_amd.train import TrainStep
    torch.manual_seed(7 + rank)                       # different noise and different samples per rank
    m = VideoCompressor()
    synth.fill_parameters(m)
    m = m.cuda()
    gop = synth.make_gop(500 + rank, 7, 64, 64).float()
    x = gop[3:4].cuda()
    refs = torch.stack([gop[0], gop[0], gop[1], gop[2]]).unsqueeze(0).cuda()
    step = TrainStep(m, train_lambda=2048.0, lr=1e-4, loss_scale=128.0)
    started = []
    orig = step.buckets.node_done

    def spy(k):
        orig(k)
        started.append(len(step.buckets._works))
    step.buckets.node_done = spy
    logs = [step(x, refs) for _ in range(3)]
    flat = torch.cat([p.detach().reshape(-1).float().cpu() for p in step.main_params])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered[1:])
    gn = torch.tensor([logs[-1]["grad_norm"]], dtype=torch.float64)
    gns = [torch.zeros_like(gn) for _ in range(world)]
    dist.all_gather(gns, gn)
    if rank == 0:
        q.put(dict(same=same, grad_norms=[float(g) for g in gns], in_sweep=max(started) if started else 0,
                   nbuckets=len(step.buckets.buckets), ready=step.buckets.ready_at, finite=all(l["rd_loss"] == l["rd_loss"] for l in logs)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_step_overlapped_all_reduce(report):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = q.get(timeout=600)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    report(f"2-rank train step: parameters identical after 3 steps: {res['same']}; grad norms {res['grad_norms']}; "
           f"{res['in_sweep']} of {res['nbuckets']} bucket all-reduces started inside the backward sweep (ready_at {res['ready']})")
    assert res["same"] and res["finite"]
    assert res["grad_norms"][0] == res["grad_norms"][1]           # the clipping norm is taken after the mean
    assert res["in_sweep"] >= 1


def _nccl_worker(port, q):
    """ONE rank, backend nccl (= RCCL): the bucket all-reduces are started from the tape hook while weight-gradient kernels
    run on the side stream.  gloo copies after a host synchronisation; RCCL is ordered against the CURRENT stream at call
    time only, so a missing side-stream -> current-stream dependency would let a reduction read (and, in place, overwrite)
    a bucket whose last weight gradient is still being accumulated.  With one rank the reduction is the identity: the
    parameters after three steps must equal the non-distributed run bit for bit."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from tdvc_amd import ops, synth
    from tdvc_amd.model.pnet import VideoCompressor
    from tdvc_amd.train import TrainStep
    ops.DETERMINISTIC = True                          # the DCN scatter's float atomics are the one order-dependent operator
    gop = synth.make_gop(500, 7, 128, 128).float()
    x = torch.cat([gop[3:4], gop[4:5]]).cuda()
    refs = torch.stack([torch.stack([gop[0], gop[0], gop[1], gop[2]]), torch.stack([gop[0], gop[1], gop[2], gop[3]])]).cuda()

    def run(force):
        torch.manual_seed(11)
        m = VideoCompressor()
        synth.fill_parameters(m)
        m = m.cuda()
        step = TrainStep(m, train_lambda=2048.0, lr=1e-4, loss_scale=128.0)
        step.buckets.force_async = force
        started = []
        orig = step.buckets.node_done

        def spy(k):
            orig(k)
            started.append(len(step.buckets._works))
        step.buckets.node_done = spy
        logs = [step(x, refs) for _ in range(3)]
        torch.cuda.synchronize()
        return torch.cat([p.detach().reshape(-1).float().cpu() for p in step.main_params]), logs, (max(started) if started else 0), len(step.buckets.buckets)

    p_plain, l_plain, _, _ = run(False)
    p_nccl, l_nccl, in_sweep, nb = run(True)
    p_again, _, _, _ = run(False)
    q.put(dict(equal=bool(torch.equal(p_plain, p_nccl)), repro=bool(torch.equal(p_plain, p_again)), in_sweep=in_sweep, nbuckets=nb,
               maxdiff=float((p_plain - p_nccl).abs().max()), loss=[l["rd_loss"] for l in l_plain], loss_nccl=[l["rd_loss"] for l in l_nccl]))
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_rccl_async_all_reduce_stream_order(report):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=900)
    p.join(timeout=120)
    assert p.exitcode == 0
    report(f"1-rank RCCL train steps: parameters bit-equal to the non-distributed run: {res['equal']} (max |diff| {res['maxdiff']:.3g}); "
           f"non-distributed run reproducible: {res['repro']}; {res['in_sweep']} of {res['nbuckets']} bucket all-reduces started inside "
           f"the backward sweep; rd_loss {res['loss']} vs {res['loss_nccl']}")
    assert res["repro"], "the deterministic training step is not reproducible run to run"
    assert res["in_sweep"] >= 1, "no all-reduce was started from the tape hook"
    assert res["equal"], "asynchronous RCCL all-reduces changed the result: a stream dependency is missing"


def test_bench_train_mode_under_one_rank_rccl_group(report):
    """`bench.py --mode train` itself (BASELINE configs[2] / [3]'s line) under a ONE-rank RCCL process group: the launcher
    contract (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*), init_process_group("nccl"), the all-gather of the rank identities
    (`ranks_seen`: device, UUID / PCI id, RCCL version -- what lets the first N > 1 run prove that RCCL saw N devices), the
    barrier / max-over-ranks timing and the JSON line, on one GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               TDVC_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "train", "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--train-batch", "2"], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    seen = line["ranks_seen"]
    report(f"bench.py --mode train under a 1-rank nccl group: {line['ms_per_step']} ms/step at batch 2; ranks_seen {seen}")
    assert line["n_gpus"] == 1 and line["value"] > 0 and len(seen) == 1
    assert seen[0]["rank"] == 0 and seen[0]["rccl"] and (seen[0]["uuid"] or seen[0]["pci"])


def test_bench_inference_mode_under_one_rank_rccl_group(report):
    """the default (inference, GOP-sharded) mode of `bench.py` under a ONE-rank RCCL process group: what the driver's N > 1 scaling runs
    execute -- nccl init with the rank's device, GOP seeds per rank, barrier / max-over-ranks timing, `ranks_seen` -- minus the legs that
    only rank 0 of a one-GPU run adds (`--no-extras --no-pmc --no-cpu-baseline`)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               TDVC_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-extras", "--no-pmc",
                        "--no-cpu-baseline"], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    report(f"bench.py (inference) under a 1-rank nccl group: {line['value']} frames/s, {line['ms_per_step']} ms/frame; ranks_seen {line['ranks_seen']}")
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["value"] > 0 and line["scaling"] == "weak"
    assert len(line["ranks_seen"]) == 1 and line["ranks_seen"][0]["rccl"]
