"""Validation metrics of the reference's MimicryLogger callback (code/glow_pytorch/mimicry_logger.py:154-251), SURVEY.md
par. 8f row 2: jerk of generated vs ground-truth motion, the invertibility check, the mismatched-context NLL probes and the
scale histograms. Every number is one more call into the engine (SeqGlow.forward / inference / invert) or a device
reduction (lfi_jerk_mean); nothing is computed on the host. Rendering (HTTP POST to the FLAME render server, :60-133) is a
separate service and stays out: `render_hook`, if set, receives the de-standardisation inputs instead.

Same hook name and arguments as the Lightning callback, so it can be registered with pytorch_lightning where that exists;
lets_face_it_amd.trainer.Trainer calls it directly after each validation batch.
"""
import random

import torch

from .glow.utils import calc_jerk, derange_batch, get_longest_history
from .glow import models, modules


class MimicryLogger:
    def __init__(self, render_hook=None, seed=None):
        self.render_hook = render_hook  # callable(name, sequence, sequence2, pl_module) or None
        # which sample gets rendered is drawn from a generator of the logger's OWN: under data parallelism the callbacks run on
        # rank 0 only, while Python's global `random` decides the negative-example branch of every training step and has to stay
        # in lockstep on all ranks (lets_face_it_glow.py:40-45; Trainer.fit)
        self._rng = random.Random(seed)

    # ------------------------------------------------------------------ mimicry_logger.py:134-152
    def log_scales(self, pl_module):
        logger = getattr(pl_module, "logger", None)
        stats = {}
        for name, x in pl_module.named_modules():
            if isinstance(x, modules.ActNorm2d):
                stats["ActNorm/" + name] = torch.exp(x.logs)
            elif isinstance(x, models.FlowStep) and x.scale is not None:
                stats["FlowStepScale/" + name] = x.scale
            elif isinstance(x, modules.InvertibleConv1x1) and pl_module.hparams.Glow["LU_decomposed"]:
                stats["InvertibleConv1x1_exp_log_s/" + name] = torch.exp(x.log_s)
        if logger is not None and hasattr(logger, "experiment") and hasattr(logger.experiment, "add_histogram"):
            for name, v in stats.items():
                logger.experiment.add_histogram(name, v.detach().cpu(), pl_module.global_step)
        return stats

    # ------------------------------------------------------------------ mimicry_logger.py:154-239
    def on_validation_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx=0):
        if batch_idx != 0:
            return
        hp = pl_module.hparams
        new_batch = {k: v.float().contiguous() for k, v in batch.items()}
        with torch.no_grad():
            z_seq, loss, _ = pl_module.seq_glow(new_batch)
            if hp.Validation["inference"]:
                seq_len = hp.Validation["seq_len"]
                cond_data = {"p1_face": new_batch["p1_face"][:, :get_longest_history(hp.Conditioning)].contiguous()}
                for k in ("p2_face", "p1_speech", "p2_speech", "frame_nb"):
                    if new_batch.get(k) is not None:
                        cond_data[k] = new_batch[k]
                predicted_seq = pl_module.seq_glow.inference(seq_len, data=cond_data)
                gt_mean_jerk = calc_jerk(new_batch["p1_face"][:, -predicted_seq.shape[1]:])
                generated_mean_jerk = calc_jerk(predicted_seq)
                pl_module.log("jerk/gt_mean", gt_mean_jerk)
                pl_module.log("jerk/generated_mean", generated_mean_jerk)
                pl_module.log("jerk/generated_mean_ratio", generated_mean_jerk / gt_mean_jerk)
                if hp.Validation["render"] and self.render_hook is not None:
                    idx = self._rng.randint(0, cond_data["p1_face"].shape[0] - 1)
                    self.render_hook("video", new_batch.get("p2_face", predicted_seq)[idx][-predicted_seq.shape[1]:],
                                     predicted_seq[idx], pl_module)
            if hp.Validation["check_invertion"]:
                pl_module.log("reconstruction/error_percentage", self.test_invertability(z_seq, loss, new_batch, pl_module))
            if hp.Validation["scale_logging"]:
                self.log_scales(pl_module)
            if hp.Validation["wrong_context_test"]:   # is the flow listening to the other modalities? (:200-239)
                mismatch = hp.Mismatch
                pl_module.log("mismatched_nll/actual_nll", loss)
                probes = []
                for kind, shuffle_time in (("shuffle_batch", False), ("shuffle_time", True)):
                    for key, mods in mismatch[kind].items():
                        if all(hp.Conditioning[x]["history"] > 0 for x in mods):
                            probes.append(("%s_%s" % (kind, key), derange_batch(new_batch, mods, shuffle_time=shuffle_time)))
                # the reference runs one forward per probe (10 at final_model.yaml); samples are independent in eval mode, so the
                # deranged batches go through the engine STACKED along the batch axis (SURVEY.md par. 8 f2: "one batched forward
                # over stacked deranged batches") and every probe's NLL is the mean over its own columns of the per-frame NLL
                for (name, _), wrong_nll in zip(probes, self.stacked_nll(pl_module, [b for _, b in probes])):
                    pl_module.log("mismatched_nll/" + name, wrong_nll)
                    pl_module.log("mismatched_nll_ratios/" + name, loss - wrong_nll)

    # frames (batch x timesteps) one stacked forward may hold: 4 x the training step's 256 x 80 (its workspaces are ~0.7 MB
    # per frame at final widths: ~40 GB of the 288 GB)
    max_stack_frames = 4 * 256 * 80

    def stacked_nll(self, pl_module, batches):
        """[(1,) loss tensor per batch dict] = SeqGlow.forward's loss of each (eval mode, no dropout), from as few engine
        forwards as `max_stack_frames` allows (all batches have the same shapes)."""
        if not batches:
            return []
        sg = pl_module.seq_glow
        if sg.training or sg.injected_masks is not None:       # dropout masks are per call: keep the reference's one-by-one loop
            return [sg(b)[1] for b in batches]
        x = batches[0]["p1_face"]
        B, T = x.shape[0], x.shape[1]
        per = max(1, min(len(batches), self.max_stack_frames // max(B * T, 1)))
        out = []
        eng = sg._ensure_engine(x.device)
        for i in range(0, len(batches), per):
            group = batches[i:i + per]
            stacked = {k: torch.cat([b[k] for b in group], dim=0).contiguous() for k in group[0]}
            sg._fwd_counter += 1     # the engine's stash and `_last` are replaced: a pending autograd graph must notice
            _, nll = eng.forward(stacked, None, with_stash=False)          # (N, len(group) * B)
            out += [nll[:, j * B:(j + 1) * B].mean().reshape(1) for j in range(len(group))]
        return out

    # ------------------------------------------------------------------ mimicry_logger.py:241-251
    def test_invertability(self, z_seq, loss, data, pl_module):
        reconstr_seq, backward_loss = pl_module.seq_glow.invert(z_seq, data)
        error_percentage = (backward_loss + loss) / loss
        if pl_module.hparams.Validation["render"] and self.render_hook is not None:
            i = self._rng.randint(0, data["p1_face"].shape[0] - 1)
            seq = torch.stack(reconstr_seq, dim=1).type_as(data["p1_face"])[i]
            self.render_hook("test_reconstr", data["p1_face"][i, -len(z_seq):, :].detach(), seq, pl_module)
        return torch.abs(error_percentage)
