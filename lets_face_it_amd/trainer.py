"""Minimal training loop standing in for the pytorch_lightning.Trainer the reference drives from train.py:37-38.

One process per GPU; with WORLD_SIZE > 1 the flat gradient buffer is all-reduced with RCCL
(torch.distributed backend "nccl") — ONE collective of n_params floats per step instead of DDP's per-bucket hooks —
and ActNorm's data-dependent init statistics and the mismatched-NLL switch are all-reduced so every rank holds
identical parameters and takes identical branches.
"""
import itertools
import os
import random
import time

import numpy as np
import torch
import torch.distributed as dist


class Trainer:
    def __init__(self, hparams, device=None, log_every=10, callbacks=None, checkpoint_dir=None):
        self.hparams = hparams
        # Lightning's `checkpoint_callback: true` default (final_model.yaml:120): rank 0 writes <dir>/last.ckpt after every
        # epoch (and at max_steps) - ONE file, overwritten (the reference's ModelCheckpoint keeps one per epoch).
        # None -> hparams.checkpoint_dir, else <hparams.default_root_dir or the current directory>/checkpoints (Lightning's own
        # default location); "" switches checkpointing off.
        if checkpoint_dir is None:
            checkpoint_dir = getattr(hparams, "checkpoint_dir", None) or os.path.join(
                getattr(hparams, "default_root_dir", None) or os.getcwd(), "checkpoints")
        self.checkpoint_dir = checkpoint_dir if getattr(hparams, "checkpoint_callback", True) else ""
        self.global_step = 0
        self.callbacks = list(callbacks or [])   # objects with on_validation_batch_end(...), e.g. MimicryLogger
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = torch.device(device) if device is not None else torch.device("cuda", self.local_rank)
        self.log_every = log_every
        self.max_epochs = int(getattr(hparams, "max_epochs", 1) or 1)
        self.max_steps = getattr(hparams, "max_steps", None)
        self.epoch = 0
        self.batches_into_epoch = 0   # batches of the current epoch already trained on (a max_steps checkpoint taken mid-epoch)
        self._epoch_rng = None        # torch's CPU generator state at the start of the current epoch (see fit)

    # ------------------------------------------------------------------ distributed plumbing
    def setup_distributed(self):
        if self.world_size > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            backend = "nccl" if self.device.type == "cuda" else "gloo"
            if self.device.type == "cuda":
                torch.cuda.set_device(self.device)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world_size)

    def allreduce_grads(self, flat, async_op=False):
        """Sum a (slice of the) flat gradient buffer over ranks in place; async_op=True returns the work handle."""
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)

    def allreduce_stats(self, sums):
        if self.world_size > 1:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        return self.world_size

    def sync_scalar(self, value):
        if self.world_size > 1:
            value = value.clone()
            dist.all_reduce(value, op=dist.ReduceOp.SUM)
            value /= self.world_size
        return value

    def broadcast_parameters(self, model):
        """Rank 0's initial weights everywhere (DDP's constructor broadcast)."""
        if self.world_size > 1:
            eng = model.seq_glow._ensure_engine(self.device)
            dist.broadcast(eng.params, src=0)
            dist.broadcast(eng.inv_p, src=0)
            dist.broadcast(eng.inv_sign, src=0)

    # ------------------------------------------------------------------ schedule (get_scheduler, utils.py:60-82; stepped once per epoch)
    def lr_at(self, epoch):
        """Learning rate of `epoch` under the three schedules get_scheduler builds, in closed form:
        "step"            StepLR(step_size, gamma):                lr0 * gamma^(epoch // step_size)
        "lambda"          LambdaLR(lambda e: e // val):            lr0 * (epoch // val)            (zero for the first `val` epochs)
        "multiplicative"  MultiplicativeLR(lambda e: e // val):    lr0 * prod_{i=1..epoch} (i // val)   (zero from epoch 1 on while
                          val > 1 - that is what the reference's lambda1 yields, utils.py:60-61,74-77; reproduced, not repaired)
        no name           the optimiser's own lr."""
        lr = float(self.hparams.lr)
        sched = self.hparams.Optim["Schedule"]
        name = sched["name"]
        if not name:
            return lr
        if name == "step":
            a = sched["args"]["step"]
            return lr * float(a["gamma"]) ** (epoch // int(a["step_size"]))
        if name == "lambda":
            return lr * (epoch // int(sched["args"]["lambda"]["val"]))
        if name == "multiplicative":
            val = int(sched["args"]["multiplicative"]["val"])
            for i in range(1, epoch + 1):
                lr *= i // val
            return lr
        raise NotImplementedError("Unimplemented Scheduler!")

    # ------------------------------------------------------------------ loop
    def fit(self, model, datamodule):
        self.setup_distributed()
        model.to(self.device)
        model.train()
        model.seq_glow.allreduce_hook = self.allreduce_stats
        model.nll_sync_hook = self.sync_scalar
        self.broadcast_parameters(model)
        allreduce = self.allreduce_grads if self.world_size > 1 else None
        # dropout masks differ per rank (each rank sees other samples), while Python's `random` — which decides the
        # negative-example branch (lets_face_it_glow.py:40-45) — stays identical on all ranks so that they take the same branch
        if self.world_size > 1:
            torch.cuda.manual_seed(torch.initial_seed() + self.rank)
            model.seq_glow.mask_seed_offset = self.rank
        step = self.global_step
        for epoch in range(self.epoch, self.max_epochs):
            self.epoch = epoch
            lr = self.lr_at(epoch)
            t0, frames = time.time(), 0
            loader = datamodule.train_dataloader()
            if hasattr(loader, "set_epoch"):
                loader.set_epoch(epoch)     # DistributedSampler.set_epoch: one shared permutation per epoch on every rank
            skip, self.batches_into_epoch = self.batches_into_epoch, 0
            # The single-process loader draws the epoch's permutation from torch's GLOBAL generator when iteration starts
            # (DataLoader(shuffle=True)). A run resumed mid-epoch has that generator at its checkpoint-time state - after the
            # permutation and the steps' own draws (derange_batch) - so the permutation is re-drawn from the state the
            # interrupted epoch started with (kept in the checkpoint), the batches already trained on are skipped, and the
            # checkpoint-time state is put back before the first new step.
            resume_rng = None
            if skip and self._epoch_rng is not None:
                resume_rng = torch.get_rng_state()
                torch.set_rng_state(self._epoch_rng)
            else:
                if skip:    # a checkpoint from before the epoch-start state was kept: the epoch's remaining length is right,
                    import warnings                                                       # its batch order is a new draw
                    warnings.warn("checkpoint holds no epoch-start RNG state: the resumed epoch is reshuffled")
                self._epoch_rng = torch.get_rng_state()
            if hasattr(loader, "skip_batches"):
                loader.skip_batches, first_bi = skip, skip      # WindowLoader starts at batch `skip` without gathering the others
            else:
                first_bi = 0
            it = iter(loader)
            head = next(it, None)                               # (draws the permutation)
            while head is not None and first_bi < skip:         # a loader without skip_batches: consume
                first_bi += 1
                head = next(it, None)
            if resume_rng is not None:
                torch.set_rng_state(resume_rng)
            self.batches_into_epoch = skip if head is not None else 0
            for bi, batch in enumerate(itertools.chain([head] if head is not None else [], it), start=skip):
                batch = {k: v.to(self.device, non_blocking=True).float().contiguous() for k, v in batch.items()}
                loss = model.fused_training_step(batch, lr, self.world_size, allreduce)
                x = batch["p1_face"]
                frames += x.shape[0] * (x.shape[1] - model.seq_glow.spec.start) * self.world_size
                step += 1
                self.global_step = step
                self.batches_into_epoch = bi + 1
                if self.rank == 0 and step % self.log_every == 0:
                    torch.cuda.synchronize()
                    print("epoch %d step %d lr %.3e loss %.4f  %.0f frames/s" %
                          (epoch, step, lr, float(loss), frames / (time.time() - t0)), flush=True)
                if self.max_steps and step >= int(self.max_steps):
                    self._checkpoint(model)
                    return
            self.validate(model, datamodule)
            self.epoch = epoch + 1          # a checkpoint written now resumes with the next epoch
            self.batches_into_epoch = 0
            self._checkpoint(model)

    def _checkpoint(self, model):
        if self.checkpoint_dir and self.rank == 0:
            os.makedirs(self.checkpoint_dir, exist_ok=True)
            self.save_checkpoint(model, os.path.join(self.checkpoint_dir, "last.ckpt"))

    def validate(self, model, datamodule):
        loader = getattr(datamodule, "val_dataloader", None)
        if loader is None:
            return None
        model.eval()
        # under data parallelism the loader hands every rank ITS share of the validation windows (WindowLoader: one shared order,
        # rank r takes elements r, r + world, ...); the loss is summed over ranks below. The callbacks (MimicryLogger: sampling,
        # invertibility, the mismatched-NLL probes, rendering) issue no collective and run on rank 0 only, on rank 0's first batch
        total = torch.zeros(2, dtype=torch.float64, device=self.device)
        with torch.no_grad():
            for i, batch in enumerate(loader()):
                batch = {k: v.to(self.device).float().contiguous() for k, v in batch.items()}
                out = model.validation_step(batch, i)
                if self.rank == 0:
                    for cb in self.callbacks:   # Lightning's hook order: after each validation batch (mimicry_logger.py:154)
                        cb.on_validation_batch_end(self, model, out, batch, i, 0)
                total[0] += out.reshape(()).double()
                total[1] += 1
        if self.world_size > 1:
            dist.all_reduce(total, op=dist.ReduceOp.SUM)
        model.train()
        val = float(total[0]) / max(float(total[1]), 1.0)
        if self.rank == 0:
            print("epoch %d val_loss %.4f" % (self.epoch, val), flush=True)
        return val

    def save_checkpoint(self, model, path):
        """Weights (state_dict incl. the `last_missmatched_nll` buffer), hparams, epoch / step counters and the fused
        optimiser's state (Adam moments over the flat parameter buffer + step count: with betas[1] = 0.9999 a resumed run
        that restarted them would take ~10^4 steps to get its second moments back)."""
        if self.rank == 0:
            eng = model.seq_glow.engine
            opt = None
            if eng is not None:
                opt = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in eng.optimizer_state().items()}
            tmp = path + ".tmp"
            torch.save({"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                        "hparams": vars(self.hparams), "epoch": self.epoch, "global_step": self.global_step,
                        "optimizer_state": opt, "actnorm_inited": bool(model.seq_glow.glow.actnorm_inited()),
                        # the negative-example branch draws from Python's `random`, derange_batch from torch's CPU generator,
                        # the loader's shuffle too: a resumed run continues those streams instead of replaying them
                        "rng": {"python": random.getstate(), "numpy": np.random.get_state(), "torch": torch.get_rng_state(),
                                # what the current epoch's shuffle was drawn from (fit re-draws it on a mid-epoch resume)
                                "torch_epoch_start": self._epoch_rng},
                        "batches_into_epoch": int(self.batches_into_epoch)}, tmp)
            os.replace(tmp, path)

    @staticmethod
    def _flat_optimizer_state(model, eng, opt_state, name="adam"):
        """A torch.optim state_dict ({'state': {i: {...}}, ...}; i = index in model.parameters()) of the optimiser `name`
        (configure_optimizers, lets_face_it_glow.py:61-72) -> the engine's flat-buffer state. Parameters are views of the flat buffer, so
        a state tensor's place is its parameter's. Adam: exp_avg / exp_avg_sq / max_exp_avg_sq (amsgrad) / step; SGD: momentum_buffer
        (torch keeps no step count for it); RMSprop: square_avg / momentum_buffer / grad_avg (centered) / step."""
        keys = {"adam": {"exp_avg": "adam_m", "exp_avg_sq": "adam_v", "max_exp_avg_sq": "opt_aux"},
                "sgd": {"momentum_buffer": "adam_m"},
                "rmsprop": {"square_avg": "adam_v", "momentum_buffer": "adam_m", "grad_avg": "opt_aux"}}[name]
        flat = {}
        step, base = 0, eng.params.data_ptr()
        for i, p in enumerate(model.parameters()):
            st = opt_state["state"].get(i)
            if st is None:
                continue
            off = (p.data_ptr() - base) // 4
            if not (0 <= off and off + p.numel() <= eng.n_params):
                raise ValueError("parameter %d does not live in the engine's flat buffer" % i)
            unknown = [k for k in st if k not in keys and k != "step"]
            if unknown:
                raise KeyError("optimizer state of parameter %d holds %s: not a torch %s state (Optim.name = %r)"
                               % (i, unknown, name, name))
            for k, dst in keys.items():
                t = st.get(k)
                if t is None:
                    continue
                if dst not in flat:
                    flat[dst] = torch.zeros_like(eng.params)
                flat[dst][off:off + p.numel()].copy_(t.reshape(-1))
            if "step" in st:
                step = max(step, int(st["step"]))
        return {"step_count": step, "optimizer": name, "momentum_inited": "adam_m" in flat and name == "sgd",
                "adam_m": flat.get("adam_m"), "adam_v": flat.get("adam_v"), "opt_aux": flat.get("opt_aux")}

    @classmethod
    def _flat_adam_state(cls, model, eng, opt_state):
        return cls._flat_optimizer_state(model, eng, opt_state, "adam")

    def resume(self, model, path):
        """Continue a run from a checkpoint of save_checkpoint: weights, ActNorm's inited flag, Adam state, epoch and step."""
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        model.load_state_dict(ckpt["state_dict"])
        model.to(self.device)
        if ckpt.get("actnorm_inited", True):
            model.seq_glow.glow.set_actnorm_init(True)
        eng = model.seq_glow._ensure_engine(self.device)
        if ckpt.get("optimizer_state") is not None:
            saved = ckpt["optimizer_state"].get("optimizer")
            if saved is not None and saved != model.hparams.Optim["name"]:
                raise ValueError("checkpoint %s holds %s state, hparams.Optim.name is %r: the state buffers would be reinterpreted "
                                 "(Adam's moments as a momentum buffer / square average)" % (path, saved, model.hparams.Optim["name"]))
            eng.load_optimizer_state(ckpt["optimizer_state"])
        elif ckpt.get("optimizer_states"):
            # a Lightning / reference checkpoint: torch.optim.Adam's per-parameter state, in model.parameters() order
            # (configure_optimizers, lets_face_it_glow.py:61-72) -> the flat moments
            # (whichever of Adam / SGD / RMSprop hparams.Optim names: their state keys differ)
            eng.load_optimizer_state(self._flat_optimizer_state(model, eng, ckpt["optimizer_states"][0],
                                                                model.hparams.Optim["name"]))
        else:
            import warnings
            warnings.warn("checkpoint %s holds no optimiser state: Adam's moments restart from zero (with betas[1] = 0.9999 "
                          "the second moments need ~10^4 steps to recover)" % path)
        rng = ckpt.get("rng")
        if rng:
            random.setstate(rng["python"])
            np.random.set_state(rng["numpy"])
            torch.set_rng_state(rng["torch"])
            self._epoch_rng = rng.get("torch_epoch_start")
        self.batches_into_epoch = int(ckpt.get("batches_into_epoch", 0))
        self.epoch = int(ckpt.get("epoch", 0))
        self.global_step = int(ckpt.get("global_step", 0))
        model.global_step = self.global_step
        return model
