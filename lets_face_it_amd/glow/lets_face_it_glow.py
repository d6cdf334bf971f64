"""LetsFaceItGlow: the training wrapper of glow_pytorch/glow/lets_face_it_glow.py without PyTorch-Lightning.

Same constructor, `seq_glow` attribute, `last_missmatched_nll` buffer, training_step / validation_step /
configure_optimizers semantics (lets_face_it_glow.py:18-72). `training_step` returns a loss whose .backward() runs the
HIP backward kernels (drop-in for an external optimiser loop); `fused_training_step` is the native path the bundled
Trainer uses: forward + backward + gradient all-reduce + clip + Adam with no autograd graph and no per-parameter
kernels.
"""
import contextlib
import random

import numpy as np
import torch
import torch.nn as nn
from torch.optim import SGD, Adam, RMSprop

from .models import SeqGlow
from .utils import derange_batch, get_mismatched_modalities, get_scheduler, test_params


def _recorded_event():
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


class LetsFaceItGlow(nn.Module):
    def __init__(self, hparams, dataset_root=None, test=None):
        super().__init__()
        test_params(hparams)
        if dataset_root is not None:
            hparams.dataset_root = dataset_root
        if test is not None:
            hparams.Test = test
        self.hparams = hparams
        self.register_buffer("last_missmatched_nll", torch.tensor(np.inf))
        self.seq_glow = SeqGlow(self.hparams)
        self.missmatched_modalities, self.missmatched_nll_name = None, None
        if self.hparams.Train["use_negative_nll_loss"]:
            self.missmatched_modalities, self.missmatched_nll_name = get_mismatched_modalities(self.hparams)
        self.logged = {}
        self.global_step = 0
        self.nll_sync_hook = None  # data-parallel trainer: averages the mismatched NLL over ranks
        self._mm_host = None
        self._mm_pending = None  # (pinned host scalar, event): the value of the last negative step on its way to the host
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.__dict__.update(_mm_host=None, _mm_pending=None))

    # Lightning's self.log, reduced to a dict of the latest values
    def log(self, name, value, **kwargs):
        self.logged[name] = value.detach() if torch.is_tensor(value) else value

    def _last_mm(self):
        """Host copy of the `last_missmatched_nll` buffer: reading the device tensor every step (`buffer > 0`) is a device
        synchronisation per step, which stops the host from queueing the next step's launches under the current one."""
        if self._mm_host is None:
            if self._mm_pending is not None:
                # the value left the device on a side stream right after the negative step's forward pass: wait for THAT copy only
                # (float(buffer) would wait for everything queued behind it - backward, optimiser - and leave the GPU idle
                # while the next step is issued)
                pinned, ev = self._mm_pending
                ev.synchronize()
                self._mm_host, self._mm_pending = float(pinned), None
            else:
                self._mm_host = float(self.last_missmatched_nll)
        return self._mm_host

    def _negative_branch(self):
        """lets_face_it_glow.py:40-45: only while the last mismatched NLL is > 0, with probability 0.1 (Python RNG)."""
        return bool(self.hparams.Train["use_negative_nll_loss"] and self._last_mm() > 0
                    and random.random() < 0.1 and self.missmatched_modalities)

    def training_step(self, batch, batch_idx):
        if self._negative_branch():
            deranged = derange_batch(batch, self.missmatched_modalities)
            _, loss, _ = self.seq_glow(deranged)
            self.log("Loss/missmatched_nll", -loss)
            self._store_mismatched(-loss.detach())
            loss = loss * -0.1
        else:
            _, loss, _ = self.seq_glow(batch)
        self.log("train_loss", loss)
        return loss

    def _store_mismatched(self, value):
        value = value.reshape(()).to(self.last_missmatched_nll.device)
        if self.nll_sync_hook is not None:
            value = self.nll_sync_hook(value)
        self.last_missmatched_nll.copy_(value)
        self._mm_host = None  # re-read (one wait, after negative steps only)
        self._mm_pending = None
        if value.is_cuda:
            main = torch.cuda.current_stream(value.device)
            side = getattr(self, "_mm_stream", None)
            if side is None or side.device != value.device:
                side = self._mm_stream = torch.cuda.Stream(device=value.device)
            pinned = torch.empty((), dtype=value.dtype).pin_memory()
            side.wait_stream(main)                    # the value is complete; nothing enqueued on `main` after this is waited for
            with torch.cuda.stream(side):
                pinned.copy_(self.last_missmatched_nll, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            self._mm_pending = (pinned, ev)

    # ------------------------------------------------------------------ the step as a hipGraph (opt-in: step_graph / LFI_STEP_GRAPH=1)
    # One optimiser step is ~100 kernel launches issued through Python + ctypes: 1.4-1.6 ms of host time per 8.6 ms step. With
    # step_graph on, from the third step of a (shape, branch) a single-GPU step is captured ONCE - dropout masks, forward,
    # backward, grad norm, Adam, on both of the engine's streams - and replayed: the host then issues three launches per step (the
    # batch copy into the graph's input buffers, lfi_set_step_params, the replay). What differs between steps lives in device
    # memory: the dropout key and Adam's bias-corrected step size (include/lfi.h, lfi_set_step_params); the ordinary and the
    # negative-example step (loss x -0.1, lets_face_it_glow.py:40-50) are two graphs. Parameters after replayed steps are
    # bit-identical to eager steps' (tests/test_gpu_headline_parity.py). Measured on MI355X / ROCm 7.0: host issue 1.5 -> 0.7-1.0 ms
    # per step (hipGraphLaunch of ~100 nodes on two streams is not free). GPU step time: within +-0.1 ms of the eager step's on the
    # builder's boxes (round 4: 6.89 against 6.95 ms), but 0.7 ms (9 %) SLOWER than eager on the driver's box of round 3 (8.40
    # against 7.71) - the eager queue is already kept full by the host running ahead, so replay has little to win and, box by
    # box, something to lose.
    # OFF by default; bench.py reports both. Data-parallel steps replay TWO graphs with the collectives between them
    # (_capture_dp_step). Always eager: injected masks, ActNorm's data-dependent init, per-kernel timing (HIP events cannot bracket a
    # kernel inside a replay), LFI_DP_SYNC=1 and the bench's per-bucket event profile.
    @contextlib.contextmanager
    def _off_default_stream(self, dev):
        """Replays never run on the legacy default stream: with GPU_MAX_HW_QUEUES=3 (what a process that trains AND samples wants,
        profiles/round6_sampler_stream_pick.md; bench.py sets it) hipGraphLaunch of the two-stream step graph into the NULL stream
        segfaults inside the HIP runtime on ROCm 7.2 (reproduced three times in a row; the same graph on any other stream, or with four
        queues, runs). When the caller is on the default stream the replay - and, under data parallelism, the collectives and clip + Adam
        between and behind the two replays - runs on a stream of this module's, joined to the caller's on both sides.
        Yields the caller's stream (None when no switch was needed)."""
        cur = torch.cuda.current_stream(dev)
        if cur != torch.cuda.default_stream(dev):
            yield None
            return
        own = self.__dict__.get("_graph_stream")
        if own is None or own.device != torch.device(dev):
            own = self._graph_stream = torch.cuda.Stream(device=dev)
        own.wait_stream(cur)
        with torch.cuda.stream(own):
            yield cur
        cur.wait_stream(own)

    def _graph_key(self, batch, negative, eng):
        return (bool(negative), eng.precision, str(eng.backward_products), tuple(sorted(eng.pass_skip.items())),
                tuple((k, tuple(v.shape)) for k, v in sorted(batch.items())))

    def _graph_allowed(self, sg, eng, world_size, allreduce):
        import os
        on = self.__dict__.get("step_graph")
        if on is None:
            on = os.environ.get("LFI_STEP_GRAPH", "0") == "1"
        import os as _os
        dp = allreduce is not None and world_size > 1
        return (bool(on) and (not dp or (_os.environ.get("LFI_DP_SYNC") != "1" and self.__dict__.get("dp_profile") is None))
                and sg.injected_masks is None and sg.training and sg.glow.actnorm_inited() and eng.timers is None
                and self.hparams.Optim["name"] == "adam")

    # Data parallelism (round 6): the collectives stay outside any graph, so the step is TWO graphs split where
    # the eager step launches the flow bucket's all-reduce (engine.backward's after_flow point: every flow gradient enqueued, the
    # window encoders' BPTT not yet): A = dropout masks + forward + the flow's backward, B = the encoders' backward; between them the
    # asynchronous all-reduce of the flow bucket, after B the encoder bucket's, then clip + Adam as three eager launches (their
    # gradient multiplier 1 / world is a kernel argument). Per step the host issues two replays, two collectives and ~6 launches
    # instead of ~100. Same kernels on the same buffers in the same order: parameters bit-identical to the eager data-parallel step
    # (tools/dp_gloo_check.py --graph; tests/test_a_gpu_dp.py).
    def _capture_dp_step(self, key, batch, negative, eng):
        sg = self.seq_glow
        dev = batch["p1_face"].device
        st = {"in": {k: torch.empty_like(v) for k, v in batch.items()},
              "params": torch.zeros(4, dtype=torch.int64, device=dev), "graph": torch.cuda.CUDAGraph(), "graph_b": torch.cuda.CUDAGraph(),
              "dp": True}
        x = batch["p1_face"]
        B, N = x.shape[0], x.shape[1] - sg.spec.start
        sign = -0.1 if negative else 1.0
        ga, gb = st["graph"], st["graph_b"]
        pool = torch.cuda.graph_pool_handle()
        cap = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize(dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        state = {"open": None}

        def split():
            ga.capture_end()
            state["open"] = None
            gb.capture_begin(pool=pool)
            state["open"] = gb

        try:
            with torch.cuda.stream(cap):
                ga.capture_begin(pool=pool)
                state["open"] = ga
                masks = eng.draw_masks(B, N, 0, key_dev=st["params"]) if eng.has_dropout() else None
                _, nll = eng.forward(st["in"], masks, with_stash=True)
                st["mean"] = nll.mean().reshape(1)
                st["numel"] = nll.numel()
                eng.backward(sign / nll.numel(), after_flow=split)
                if state["open"] is not gb:
                    raise RuntimeError("engine.backward did not reach its after_flow point")
                gb.capture_end()
                state["open"] = None
        except Exception:
            if state["open"] is not None:      # leave capture mode before the error travels on
                try:
                    state["open"].capture_end()
                except Exception:      # noqa: BLE001
                    pass
            raise
        finally:
            torch.cuda.current_stream(dev).wait_stream(cap)
        st["dropout"] = masks is not None
        return st

    def _replay_dp_step(self, st, batch, lr, negative, eng, world_size, allreduce):
        from .. import _lib
        sg = self.seq_glow
        for k, v in st["in"].items():
            v.copy_(batch[k], non_blocking=True)
        a = self.hparams.Optim["args"]["adam"]
        step_size, inv_sqrt_bc2 = eng.adam_step_floats(lr, float(a["betas"][0]), float(a["betas"][1]), eng.step_count + 1)
        seed = (torch.initial_seed() + sg.mask_seed_offset) & (2 ** 64 - 1)
        _lib.check(eng.L.lfi_set_step_params(st["params"].data_ptr(), seed, eng._mask_calls + 1, step_size, inv_sqrt_bc2,
                                             torch.cuda.current_stream().cuda_stream), "lfi_set_step_params")
        off = eng.flow_offset
        st["graph"].replay()                                     # masks, forward, the flow's backward
        pending = allreduce(eng.grads[off:], async_op=True)      # travels under graph B
        st["graph_b"].replay()                                   # the window encoders' backward
        if off > 0:
            allreduce(eng.grads[:off])
        if pending is not None:
            pending.wait()
        clip = float(getattr(self.hparams, "gradient_clip_val", 0) or 0)
        eng.optimizer_step(0.0, float(a["betas"][0]), float(a["betas"][1]), float(a["eps"]), clip=clip, gmul=1.0 / world_size,
                           hyper_dev=st["params"].data_ptr() + 16, weight_decay=float(a.get("weight_decay", 0) or 0),
                           amsgrad=bool(a.get("amsgrad", False)))
        eng.step_count += 1
        if st["dropout"]:
            eng._mask_calls += 1
        sg._fwd_counter += 1
        mean = st["mean"].clone()
        if negative:
            self.log("Loss/missmatched_nll", -mean)
            self._store_mismatched(-mean)
            mean = mean * -0.1
        self.global_step += 1
        self.log("train_loss", mean)
        return mean.detach()

    def _capture_step(self, key, batch, negative, eng):
        from .. import _lib
        sg = self.seq_glow
        dev = batch["p1_face"].device
        st = {"in": {k: torch.empty_like(v) for k, v in batch.items()},
              "params": torch.zeros(4, dtype=torch.int64, device=dev), "graph": torch.cuda.CUDAGraph()}
        x = batch["p1_face"]
        B, N = x.shape[0], x.shape[1] - sg.spec.start
        a = self.hparams.Optim["args"]["adam"]
        clip = float(getattr(self.hparams, "gradient_clip_val", 0) or 0)
        sign = -0.1 if negative else 1.0
        torch.cuda.synchronize(dev)
        with torch.cuda.graph(st["graph"]):
            masks = eng.draw_masks(B, N, 0, key_dev=st["params"]) if eng.has_dropout() else None
            _, nll = eng.forward(st["in"], masks, with_stash=True)
            st["mean"] = nll.mean().reshape(1)
            eng.backward(sign / nll.numel())
            eng.optimizer_step(0.0, float(a["betas"][0]), float(a["betas"][1]), float(a["eps"]), clip=clip, gmul=1.0,
                               hyper_dev=st["params"].data_ptr() + 16, weight_decay=float(a.get("weight_decay", 0) or 0),
                               amsgrad=bool(a.get("amsgrad", False)))
        st["dropout"] = masks is not None
        return st

    def _replay_step(self, st, batch, lr, negative, eng):
        from .. import _lib
        sg = self.seq_glow
        for k, v in st["in"].items():
            v.copy_(batch[k], non_blocking=True)
        a = self.hparams.Optim["args"]["adam"]
        step_size, inv_sqrt_bc2 = eng.adam_step_floats(lr, float(a["betas"][0]), float(a["betas"][1]), eng.step_count + 1)
        seed = (torch.initial_seed() + sg.mask_seed_offset) & (2 ** 64 - 1)
        _lib.check(eng.L.lfi_set_step_params(st["params"].data_ptr(), seed, eng._mask_calls + 1, step_size, inv_sqrt_bc2,
                                             torch.cuda.current_stream().cuda_stream), "lfi_set_step_params")
        st["graph"].replay()
        eng.step_count += 1
        if st["dropout"]:
            eng._mask_calls += 1
        sg._fwd_counter += 1
        mean = st["mean"].clone()          # the graph's own output buffer is rewritten by the next replay
        if negative:
            self.log("Loss/missmatched_nll", -mean)
            self._store_mismatched(-mean)
            mean = mean * -0.1
        self.global_step += 1
        self.log("train_loss", mean)
        return mean.detach()

    def fused_training_step(self, batch, lr, world_size=1, allreduce=None):
        """One optimiser step entirely in the engine. Returns the (detached) loss of this rank.

        allreduce: callable(fp32 gradient tensor, async_op=False) summing it over ranks in place (RCCL) and returning
        the torch.distributed work handle when async_op, or None.
        """
        sg = self.seq_glow
        negative = self._negative_branch()
        if negative:
            batch = derange_batch(batch, self.missmatched_modalities)
        x = batch["p1_face"]
        eng = sg._ensure_engine(x.device)
        B, T = x.shape[0], x.shape[1]
        N = T - sg.spec.start
        if self._graph_allowed(sg, eng, world_size, allreduce):
            graphs = self.__dict__.setdefault("_step_graphs", {})
            if graphs.get("engine") is not eng:      # a re-bound engine (.to() / .float()): its buffers are gone
                graphs.clear()
                graphs["engine"] = eng
            key = self._graph_key(batch, negative, eng)
            dp = allreduce is not None and world_size > 1
            capture = self._capture_dp_step if dp else self._capture_step
            if dp:
                key = ("dp",) + key
            st = graphs.get(key)
            if st is None and graphs.get(("seen",) + key, 0) >= 2 and not graphs.get("broken"):
                try:
                    st = graphs[key] = capture(key, batch, negative, eng)
                    # the other branch's graph now too (same shapes, same kernels; capturing executes nothing): a negative step
                    # turns up once in ten steps, and its capture should not land in the middle of somebody's timed region
                    if self.hparams.Train["use_negative_nll_loss"] and self.missmatched_modalities:
                        other = (("dp",) if dp else ()) + self._graph_key(batch, not negative, eng)
                        if other not in graphs:
                            graphs[other] = capture(other, batch, not negative, eng)
                except Exception as e:   # a runtime that cannot capture this step keeps launching it eagerly
                    import warnings
                    warnings.warn("hipGraph capture of the training step failed (%s: %s); staying with eager launches"
                                  % (type(e).__name__, e))
                    graphs["broken"] = True
                    torch.cuda.synchronize(x.device)
            if st is not None:
                with self._off_default_stream(x.device) as back:
                    if st.get("dp"):
                        out = self._replay_dp_step(st, batch, lr, negative, eng, world_size, allreduce)
                    else:
                        out = self._replay_step(st, batch, lr, negative, eng)
                    if back is not None:
                        out.record_stream(back)
                return out
            graphs[("seen",) + key] = graphs.get(("seen",) + key, 0) + 1
        masks = sg._draw_masks(B, N, x.device)
        init = sg._allreduce() if (sg.training and not sg.glow.actnorm_inited()) else None
        sg._fwd_counter += 1
        _, nll = eng.forward(batch, masks, with_stash=True, init_actnorm=init)
        if init is not None:
            sg.glow.set_actnorm_init(True)
        loss = nll.mean().reshape(1)
        sign = 1.0
        if negative:
            self.log("Loss/missmatched_nll", -loss)
            self._store_mismatched(-loss)
            sign = -0.1
            loss = loss * -0.1
        if allreduce is not None and world_size > 1:
            # two buckets: the flow block (>95 % of the floats) is all-reduced asynchronously as soon as its gradients are
            # enqueued and travels over xGMI under the window encoders' BPTT; the encoder block follows on the main stream.
            # LFI_DP_SYNC=1: both buckets synchronously after the whole backward pass (what the overlap buys = the difference).
            # dp_profile (a list, bench.py): per step the HIP events that say how long each bucket held the main stream.
            import os
            off = eng.flow_offset
            prof = self.__dict__.get("dp_profile")
            mark = (lambda: _recorded_event()) if prof is not None else (lambda: None)
            if os.environ.get("LFI_DP_SYNC") == "1":
                eng.backward(sign / nll.numel())
                e0 = mark()
                allreduce(eng.grads[off:])
                e1 = mark()
                if off > 0:
                    allreduce(eng.grads[:off])
                e2 = mark()
                if prof is not None:
                    prof.append({"mode": "sync", "flow_bucket": (e0, e1), "encoder_bucket": (e1, e2)})
            else:
                pending, ev = [], {}

                def launch_flow_bucket():
                    ev["flow_launched"] = mark()
                    pending.append(allreduce(eng.grads[off:], async_op=True))

                eng.backward(sign / nll.numel(), after_flow=launch_flow_bucket)
                e1 = mark()        # the window encoders' BPTT is enqueued behind this point's predecessors
                if off > 0:
                    allreduce(eng.grads[:off])
                e2 = mark()
                for work in pending:
                    if work is not None:
                        work.wait()
                e3 = mark()
                if prof is not None:
                    # bptt_window: main-stream time between the flow bucket's launch and the end of the encoder BPTT (what the
                    # asynchronous bucket can hide under); flow_bucket_exposed: how long the main stream then still waited for it
                    # (~0 = it finished before the BPTT and the encoder bucket did)
                    prof.append({"mode": "overlap", "bptt_window": (ev.get("flow_launched"), e1), "encoder_bucket": (e1, e2),
                                 "flow_bucket_exposed": (e2, e3)})
        else:
            eng.backward(sign / nll.numel())
        self._fused_optimizer_step(eng, lr, 1.0 / world_size)
        self.global_step += 1
        self.log("train_loss", loss)
        return loss.detach()

    def _fused_optimizer_step(self, eng, lr, gmul):
        """configure_optimizers' three choices (lets_face_it_glow.py:61-72: Adam / SGD / RMSprop built as `cls(params, lr=lr,
        **Optim["args"][name])`, torch defaults for whatever the YAML leaves out) as one launch on the flat buffers, the Trainer's
        gradient_clip_val in front. Unknown keyword arguments raise as torch's constructors would."""
        opt = self.hparams.Optim
        name = opt["name"]
        a = dict((opt.get("args") or {}).get(name) or {})
        clip = float(getattr(self.hparams, "gradient_clip_val", 0) or 0)
        if name == "adam":
            betas = a.pop("betas", (0.9, 0.999))
            eps = float(a.pop("eps", 1e-8))
            wd = float(a.pop("weight_decay", 0) or 0)
            ams = bool(a.pop("amsgrad", False))
            if wd < 0:
                raise ValueError("Invalid weight_decay value: %s" % wd)      # torch.optim.Adam's own check
            if a:
                raise TypeError("Adam got unexpected arguments %s" % sorted(a))
            eng.optimizer_step(lr, float(betas[0]), float(betas[1]), eps, clip=clip, gmul=gmul, weight_decay=wd, amsgrad=ams)
        elif name == "sgd":
            kw = {k: a.pop(k) for k in ("momentum", "dampening", "weight_decay", "nesterov") if k in a}
            if a:
                raise TypeError("SGD got unexpected arguments %s" % sorted(a))
            eng.optimizer_step_sgd(lr, clip=clip, gmul=gmul, **kw)
        elif name == "rmsprop":
            kw = {k: a.pop(k) for k in ("alpha", "eps", "weight_decay", "momentum", "centered") if k in a}
            if a:
                raise TypeError("RMSprop got unexpected arguments %s" % sorted(a))
            eng.optimizer_step_rmsprop(lr, clip=clip, gmul=gmul, **kw)
        else:
            raise KeyError(name)

    def validation_step(self, batch, batch_idx):
        with torch.no_grad():
            _, loss, _ = self.seq_glow(batch)
        self.log("val_loss", loss)
        return loss

    def configure_optimizers(self):
        name = self.hparams.Optim["name"]
        optimizer = {"adam": Adam, "sgd": SGD, "rmsprop": RMSprop}[name](
            self.parameters(), lr=self.hparams.lr, **self.hparams.Optim["args"][name])
        return [optimizer], get_scheduler(self.hparams.Optim["Schedule"], optimizer)

    @classmethod
    def load_from_checkpoint(cls, path, dataset_root=None, test=None, map_location="cpu"):
        """Accepts a Lightning checkpoint of the reference ({'state_dict', 'hyper_parameters' | 'hparams'}) or one
        written by lets_face_it_amd.trainer.Trainer.save_checkpoint."""
        from argparse import Namespace
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
        hp = ckpt.get("hyper_parameters") or ckpt.get("hparams")
        if isinstance(hp, dict):
            hp = Namespace(**hp)
        model = cls(hp, dataset_root=dataset_root, test=test)
        model.load_state_dict(ckpt["state_dict"])
        model.seq_glow.glow.set_actnorm_init(True)  # a trained checkpoint must not re-initialise ActNorm
        return model
