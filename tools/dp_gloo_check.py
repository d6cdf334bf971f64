"""Two (or more) ranks on ONE GPU over gloo: the engine's data-parallel step. Launch with
  timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/dp_gloo_check.py
Every rank prints a checksum of its parameters after ActNorm init + 4 Adam steps; rank 0 also steps ONE process' worth of the
concatenated batch and prints the relative difference."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402



GRAPH = "--graph" in sys.argv   # the data-parallel step as two replayed hipGraphs with the collectives between them vs eager launches
NCCL1 = "--nccl1" in sys.argv   # ONE rank on backend "nccl" (= RCCL): the collectives' launch path on this ROCm, no transport
FINAL = "--final" in sys.argv or NCCL1   # final_model.yaml widths at BASELINE's synthetic dims, batch 256 per rank, T = 80


class _Dims:
    def __init__(self, T, start, C, S):
        self.T, self.start, self.C, self.S = T, start, C, S


def _make(fx_name, device):
    from argparse import Namespace
    import random
    import numpy as np
    from helpers import Fixture
    from lets_face_it_amd.glow.lets_face_it_glow import LetsFaceItGlow
    if FINAL:
        from lets_face_it_amd.glow.utils import load_hparams_file
        hp = load_hparams_file(os.path.join(ROOT, "lets_face_it_amd", "hparams", "final_model_synthetic.yaml"))
        hp["gradient_clip_val"] = 20
        hp["Train"]["use_negative_nll_loss"] = False
        random.seed(0)
        np.random.seed(0)
        torch.manual_seed(0)
        m = LetsFaceItGlow(Namespace(**hp))
        g = torch.Generator().manual_seed(4321)
        with torch.no_grad():     # LinearZeros is zero at init: perturb it so that the conditioning path carries gradient
            for name, p in m.named_parameters():
                if "final_linear" in name:
                    p.add_(torch.randn(p.shape, generator=g) * 0.05)
        m.to(device).train()
        m.seq_glow.precision = "f32"
        return _Dims(80, m.seq_glow.spec.start, m.seq_glow.spec.C, m.seq_glow.spec.S), hp, m
    fx = Fixture(fx_name)
    hp = fx.hp
    hp["gradient_clip_val"] = 20
    hp["Train"]["use_negative_nll_loss"] = False
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    m = LetsFaceItGlow(Namespace(**hp))
    m.seq_glow.load_state_dict(fx.state_dict(torch.float32))
    m.to(device).train()
    m.seq_glow.precision = "f32"
    return fx, hp, m


def _batches(fx, steps, B):
    g = torch.Generator().manual_seed(7)
    return [{k: torch.randn(B, fx.T, d, generator=g) for k, d in
             (("p1_face", fx.C), ("p2_face", fx.C), ("p1_speech", fx.S), ("p2_speech", fx.S))} for _ in range(steps)]


def run(rank, world, dev, tr, ranks=2):
    """world > 1: this rank's shard of every batch; world == 1: ONE process on the concatenated batch of `ranks` ranks."""
    from argparse import Namespace
    fx, hp, m = _make("mid", dev)
    if tr is None and world > 1:
        from lets_face_it_amd.trainer import Trainer
        tr = Trainer(Namespace(**hp), device=dev)
        m.seq_glow.allreduce_hook = tr.allreduce_stats
        m.nll_sync_hook = tr.sync_scalar
        tr.broadcast_parameters(m)
    B = 256 if FINAL else 16
    for step, full in enumerate(_batches(fx, 3 if FINAL else 4, world * B if world > 1 else ranks * B)):
        nb = full["p1_face"].shape[0]
        lo, hi = (rank * B, (rank + 1) * B) if world > 1 else (0, nb)
        shard = {k: v[lo:hi].to(dev).contiguous() for k, v in full.items()}
        gm = torch.Generator().manual_seed(100 + step)
        masks = {}
        for e in m.seq_glow.spec.encoders:
            if e.dropout > 0:
                keep = 1.0 - e.dropout
                mk = (torch.rand(fx.T - fx.start, nb, e.hist, generator=gm) < keep).float() / keep
                masks[e.name] = mk[:, lo:hi].contiguous().to(dev)
        m.seq_glow.injected_masks = masks
        if world > 1:
            m.fused_training_step(shard, 1e-3, world, tr.allreduce_grads)
        else:
            m.fused_training_step(shard, 1e-3)
        print("rank %d step %d done" % (rank, step), flush=True)
    torch.cuda.synchronize()
    return m.seq_glow.engine.params.detach().cpu()


def run_graph_leg(rank, world, dev, graph, steps=8):
    """`steps` data-parallel optimiser steps with the engine's own dropout masks (a captured step cannot take injected ones) and the
    negative-example branch off; graph=True: LetsFaceItGlow.step_graph (from the third step of the shape on the step is two replayed
    hipGraphs with the two gradient buckets' all-reduces between and behind them). -> (parameters, Adam moments, replayed steps)"""
    from argparse import Namespace
    from lets_face_it_amd.trainer import Trainer
    fx, hp, m = _make("mid", dev)
    m.step_graph = graph
    tr = Trainer(Namespace(**hp), device=dev)
    m.seq_glow.allreduce_hook = tr.allreduce_stats
    m.nll_sync_hook = tr.sync_scalar
    tr.broadcast_parameters(m)
    B = 256 if FINAL else 16
    torch.manual_seed(11)          # the dropout masks' key is (torch.initial_seed() + offset, call counter): the same in both legs
    replayed = 0
    for step, full in enumerate(_batches(fx, steps, world * B)):      # (on the default stream, as the trainer and bench.py run)
        shard = {k: v[rank * B:(rank + 1) * B].to(dev).contiguous() for k, v in full.items()}
        m.fused_training_step(shard, 1e-3, world, tr.allreduce_grads)
        g = m.__dict__.get("_step_graphs") or {}
        replayed += 1 if any(isinstance(v, dict) and v.get("dp") for v in g.values()) else 0
        if os.environ.get("DP_CHECK_TRACE") == "1":      # (synchronises every step: diagnosis only)
            torch.cuda.synchronize()
            e = m.seq_glow.engine
            off = e.flow_offset
            if step == 0:
                mk = e._ws.get("dropout_masks")
                print("trace rank %d step 0: an_bias %.10e an_logs %.10e masks %s loss %s" % (
                    rank, float(e.fview("an_bias").double().sum()), float(e.fview("an_logs").double().sum()),
                    "none" if mk is None else "%.6e" % float(mk.double().sum()), float(m.logged.get("train_loss", torch.zeros(1)).sum())), flush=True)
            print("trace rank %d graph %d step %d: grads enc %.10e flow %.10e | params enc %.10e flow %.10e | step_count %d mask_calls %d"
                  % (rank, int(graph), step, float(e.grads[:off].double().sum()), float(e.grads[off:].double().sum()),
                     float(e.params[:off].double().sum()), float(e.params[off:].double().sum()), e.step_count, e._mask_calls), flush=True)
    torch.cuda.synchronize()
    eng = m.seq_glow.engine
    return eng.params.detach().clone(), eng.adam_m.detach().clone(), eng.adam_v.detach().clone(), replayed


def main():
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl" if NCCL1 else "gloo", rank=rank, world_size=world)
    print("rank %d of %d up (%s)" % (rank, world, dist.get_backend()), flush=True)
    if GRAPH and os.environ.get("DP_CHECK_EAGER_TWICE") == "1":      # diagnosis: is a SECOND eager model in the process the same as the first?
        pa = run_graph_leg(rank, world, dev, False, steps=2)[0]
        pb = run_graph_leg(rank, world, dev, False, steps=2)[0]
        pc = run_graph_leg(rank, world, dev, False, steps=2)[0]
        print("rank %d eager legs: %.12e %.12e %.12e" % (rank, float(pa.double().sum()), float(pb.double().sum()), float(pc.double().sum())), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0)
    if GRAPH:
        pe, me, ve, _ = run_graph_leg(rank, world, dev, False)
        pg, mg, vg, replayed = run_graph_leg(rank, world, dev, True)
        same = torch.equal(pe, pg) and torch.equal(me, mg) and torch.equal(ve, vg)
        print("rank %d checksums: eager leg %.12e, graph leg %.12e" % (rank, float(pe.double().sum()), float(pg.double().sum())), flush=True)
        if not same or replayed < 3:
            print("rank %d: eager vs graph legs differ: params %d of %d elements (max abs %.3e), adam_m %d, adam_v %d; finite %s / %s; "
                  "steps on graphs %d" % (rank, int((pe != pg).sum()), pe.numel(), float((pe - pg).abs().max()), int((me != mg).sum()),
                                          int((ve != vg).sum()), bool(torch.isfinite(pe).all()), bool(torch.isfinite(pg).all()), replayed),
                  flush=True)
        flags = torch.tensor([1.0 if same else 0.0, float(replayed)], dtype=torch.float64)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        gathered = [torch.zeros_like(pg.cpu()) for _ in range(world)]
        dist.all_gather(gathered, pg.cpu())
        ranks_same = all(torch.equal(gathered[0], g) for g in gathered)
        if rank == 0:
            print("graph data-parallel step: parameters and Adam moments bit-identical to the eager data-parallel step on every rank: %s; "
                  "steps on captured graphs (min over ranks): %d; ranks identical: %s" % (bool(flags[0] > 0), int(flags[1]), ranks_same), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if (flags[0] > 0 and flags[1] >= 3 and ranks_same) else 1)
    if NCCL1:
        # world size 1 on RCCL: fused_training_step with the trainer's real hooks (two-bucket asynchronous all-reduce on the
        # flat gradient, ActNorm statistics, parameter broadcast) against the same steps without any collective
        from argparse import Namespace
        from lets_face_it_amd.trainer import Trainer
        fx, hp, m = _make("mid", dev)
        tr = Trainer(Namespace(**hp), device=dev)
        m.seq_glow.allreduce_hook = lambda sums: (dist.all_reduce(sums), 1)[1]
        dist.broadcast(m.seq_glow._ensure_engine(dev).params, src=0)
        fx2, hp2, m2 = _make("mid", dev)
        g = torch.Generator().manual_seed(3)
        for step in range(3):
            batch = {k: torch.randn(256, fx.T, d, generator=g).to(dev) for k, d in
                     (("p1_face", fx.C), ("p2_face", fx.C), ("p1_speech", fx.S), ("p2_speech", fx.S))}
            torch.manual_seed(50 + step)
            m.fused_training_step(batch, 1e-3, 2, lambda t, async_op=False: (t.mul_(2.0), tr.allreduce_grads(t, async_op))[1])
            torch.manual_seed(50 + step)
            m2.fused_training_step(batch, 1e-3)
        torch.cuda.synchronize()
        same = torch.equal(m.seq_glow.engine.params, m2.seq_glow.engine.params)
        print("nccl world-1 rehearsal: parameters identical to the collective-free steps: %s" % same, flush=True)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if same else 1)
    p = run(rank, world, dev, None)
    print("rank %d params checksum %.9e" % (rank, float(p.double().sum())), flush=True)
    gathered = [torch.zeros_like(p) for _ in range(world)]
    dist.all_gather(gathered, p)
    if rank == 0:
        same = all(torch.equal(gathered[0], g) for g in gathered)
        ref = run(0, 1, dev, None, ranks=world)
        err = float((p - ref).abs().max() / ref.abs().max())
        print("ranks identical: %s; vs one process on the concatenated batch: max rel diff %.2e" % (same, err), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
