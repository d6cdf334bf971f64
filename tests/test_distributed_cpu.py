"""CPU, world_size 2 over gloo: the data-parallel plumbing of lets_face_it_amd.trainer and the DP arithmetic it
relies on (mean of per-rank gradients of per-rank mean losses == gradient of the global-batch mean loss; ActNorm
init statistics summed over ranks == statistics of the concatenated batch)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from argparse import Namespace
    from helpers import Fixture
    from oracle import seqglow_oracle as oracle
    from lets_face_it_amd.trainer import Trainer
    torch.set_num_threads(2)
    fx = Fixture("tiny")
    tr = Trainer(Namespace(**fx.hp), device="cpu")
    tr.setup_distributed()
    assert dist.get_world_size() == world and tr.rank == rank

    # (1) gradient all-reduce of the flat buffer + 1/world in the optimiser == global-batch gradient
    sd = {k: v.requires_grad_(v.dtype.is_floating_point and not k.endswith((".p", ".sign_s")))
          for k, v in fx.state_dict().items()}
    batch = fx.batch()
    half = fx.B // world
    shard = {k: v[rank * half:(rank + 1) * half] for k, v in batch.items()}
    oracle.seqglow_forward(fx.hp, sd, shard)[1].sum().backward()
    names = [k for k, v in sd.items() if v.requires_grad]
    flat = torch.cat([sd[k].grad.reshape(-1) for k in names])
    # the trainer's two-bucket protocol (fused_training_step): tail bucket asynchronously, head bucket, then wait
    off = flat.numel() // 20
    work = tr.allreduce_grads(flat[off:], async_op=True)
    tr.allreduce_grads(flat[:off])
    work.wait()
    flat /= world
    full = {k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in sd.items()}
    oracle.seqglow_forward(fx.hp, full, batch)[1].sum().backward()
    ref = torch.cat([full[k].grad.reshape(-1) for k in names])
    err_grad = float((flat - ref).abs().max() / ref.abs().max())

    # (2) ActNorm init statistics: sum over ranks == concatenated batch
    x = batch["p1_face"][:, fx.start]
    xs = x[rank * half:(rank + 1) * half]
    sums = torch.cat([xs.sum(0), (xs ** 2).sum(0)])
    n = tr.allreduce_stats(sums)
    err_stats = float((sums - torch.cat([x.sum(0), (x ** 2).sum(0)])).abs().max())

    # (3) scalar sync used for the mismatched-NLL switch
    v = tr.sync_scalar(torch.tensor(float(rank + 1)))

    # (4) an epoch over a window count that is NOT a multiple of world * batch: one collective per
    # optimiser step, as fused_training_step issues; unequal step counts would leave one rank waiting in all_reduce
    from lets_face_it_amd.mimicry_data_module import WindowLoader

    class Windows:
        def __len__(self):
            return 53

        def batch(self, idx):
            return {"idx": torch.as_tensor(idx).clone()}

    steps, seen = 0, []
    for epoch in range(2):
        ld = WindowLoader(Windows(), 6, shuffle=True, rank=rank, world_size=world, seed=5)
        ld.set_epoch(epoch)
        for b in ld:
            size = torch.tensor([float(b["idx"].numel())])
            dist.all_reduce(size)                       # the step's collective
            assert float(size) == world * b["idx"].numel()   # same batch size on every rank
            seen.append(b["idx"])
            steps += 1
    count = torch.tensor([float(steps)])
    dist.all_reduce(count)
    assert float(count) == world * steps == world * 2 * 5     # ceil(ceil(53 / 2) / 6) = 5 batches per rank and epoch
    # (5) the step's two collectives at the real sizes: the flow bucket (66 MB here, as at final widths)
    # is issued asynchronously and is still travelling when the encoder bucket's synchronous all-reduce is issued behind it;
    # both ranks issue them in the same order, both complete, every float summed exactly once
    big = torch.full((16_500_000,), float(rank + 1))
    small = torch.full((840_000,), float(10 * (rank + 1)))
    work = tr.allreduce_grads(big, async_op=True)
    in_flight = not work.is_completed()
    tr.allreduce_grads(small)
    work.wait()
    ok_order = bool((big == 3.0).all()) and bool((small == 30.0).all())

    # (6) Trainer.validate under data parallelism: every rank scores ITS share of the validation windows, the loss is the
    # mean over all ranks' batches, and the callbacks run on rank 0 only
    class FakeModel:
        def eval(self):
            return self

        def train(self):
            return self

        def validation_step(self, batch, i):
            return batch["idx"].float().mean().reshape(1)

    class FakeData:
        def val_dataloader(self):
            return WindowLoader(Windows(), 6, shuffle=False, rank=rank, world_size=world)

    calls = []

    class Cb:
        def on_validation_batch_end(self, *a):
            calls.append(a[4])

    tr.callbacks = [Cb()]
    val = tr.validate(FakeModel(), FakeData())
    # 53 windows wrap to 54: rank r scores r, r + 2, ... in batches of 6 -> 5 batches per rank; the expected mean of batch means
    order = torch.cat([torch.arange(53), torch.arange(1)])
    means = [order[r:54:world][b:b + 6].float().mean() for r in range(world) for b in range(0, 27, 6)]
    ok_val = abs(val - float(torch.stack(means).double().mean())) < 1e-4 and (len(calls) == 5 if rank == 0 else len(calls) == 0)
    flags = torch.tensor([float(ok_order), float(ok_val)])
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put((err_grad, err_stats, n, float(v), bool(flags[0]), bool(flags[1]), in_flight))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_data_parallel():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        err_grad, err_stats, n, v, ok_order, ok_val, in_flight = out.get(timeout=240)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    assert err_grad < 1e-10 and err_stats < 1e-10 and n == 2 and v == 1.5
    assert ok_order, "two collectives (async flow bucket, then the encoder bucket) did not both complete with every float summed once"
    assert ok_val, "sharded validation: wrong global mean, or a callback ran off rank 0"
    print("async flow-bucket all-reduce still in flight when the encoder bucket was issued: %s" % in_flight)
