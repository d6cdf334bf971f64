"""hparams loading and small helpers with the names and behaviour of glow_pytorch/glow/utils.py.

get_hparams (utils.py:13-41) merged the YAML/JSON file into PyTorch-Lightning's Trainer argparse namespace. There is
no Lightning here: the same flat keys (gpus, max_epochs, gradient_clip_val, lr, batch_size, ...) are read from the
file, and any `--key value` pair on the command line overrides the key of that name, which is what the reference's
CLI amounted to for the keys its YAML files hold.
"""
import json
import os
import sys
from argparse import ArgumentParser, Namespace

import torch
import yaml
from torch.optim.lr_scheduler import LambdaLR, MultiplicativeLR, StepLR


def _strip_json_comments(text):
    out, i, n, in_str = [], 0, len(text), False
    while i < n:
        ch = text[i]
        if in_str:
            out.append(ch)
            if ch == "\\" and i + 1 < n:
                out.append(text[i + 1])
                i += 1
            elif ch == '"':
                in_str = False
        elif ch == '"':
            in_str = True
            out.append(ch)
        elif text.startswith("//", i):
            while i < n and text[i] != "\n":
                i += 1
            continue
        elif text.startswith("/*", i):
            i = text.find("*/", i + 2)
            i = n if i < 0 else i + 2
            continue
        else:
            out.append(ch)
        i += 1
    return "".join(out)


def load_hparams_file(path, dataset_root=None):
    """The file part of get_hparams: dict with the rnn_type default applied (utils.py:23-33)."""
    with open(path) as f:
        text = f.read()
    if path.endswith(".json"):
        hp = json.loads(_strip_json_comments(text))
    elif path.endswith((".yaml", ".yml")):
        hp = yaml.load(text, Loader=yaml.FullLoader)
    else:
        raise ValueError("hparams file must be .yaml or .json: %s" % path)
    if dataset_root is not None:
        hp["dataset_root"] = str(dataset_root)
    if not hp["Glow"].get("rnn_type"):
        hp["Glow"]["rnn_type"] = "gru"
    return hp


def _coerce(text, like):
    if isinstance(like, bool):
        return text.lower() in ("1", "true", "yes")
    if isinstance(like, int) and not isinstance(like, bool):
        return int(text)
    if isinstance(like, float):
        return float(text)
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


def get_hparams(argv=None):
    """-> (Namespace, conf_name). argv defaults to sys.argv[1:]: `<hparams_file> [--key value ...]`."""
    parser = ArgumentParser()
    parser.add_argument("hparams_file")
    args, unknown = parser.parse_known_args(sys.argv[1:] if argv is None else argv)
    conf_name = os.path.basename(args.hparams_file)
    data_dir = os.environ.get("LFI_DATA_DIR", "/data")  # misc/shared.py DATA_DIR, config.toml [project] data_dir
    params = load_hparams_file(args.hparams_file, dataset_root=data_dir)
    i = 0
    while i < len(unknown):
        tok = unknown[i]
        if not tok.startswith("--"):
            raise SystemExit("unexpected argument %r" % tok)
        key, _, val = tok[2:].partition("=")
        if not val:
            if i + 1 < len(unknown) and not unknown[i + 1].startswith("--"):
                val = unknown[i + 1]
                i += 1
            else:
                val = "true"
        params[key] = _coerce(val, params.get(key))
        i += 1
    params["hparams_file"] = args.hparams_file
    return Namespace(**params), conf_name


def get_longest_history(cond_params):
    return max(cond_params[m]["history"] for m in ("p1_face", "p1_speech", "p2_speech", "p2_face"))


def calc_jerk(x):
    """Mean absolute third difference along time of a (B, T, C) sequence (utils.py:53-58), reduced on the device by
    lfi_jerk_mean (fp32 differences as the reference, fixed-order fp64 sum). Returns a 0-dim device tensor; a CPU tensor
    raises (the reference moves its input to the CPU first; here the metric stays where the sequence was generated)."""
    if not x.is_cuda:
        raise RuntimeError("calc_jerk (lets_face_it_amd) runs on the GPU only; got a %s tensor" % x.device.type)
    from .. import _lib
    x = x.detach().float().contiguous()
    B, T, C = x.shape
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    work = torch.empty(1024, dtype=torch.float64, device=x.device)
    _lib.check(_lib.lib().lfi_jerk_mean(x.data_ptr(), B, T, C, out.data_ptr(), work.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "lfi_jerk_mean")
    return out[0]


def lambda1(val):
    return lambda epoch: epoch // val


def get_scheduler(sched_params, optimizer):
    name = sched_params["name"]
    if not name:
        return optimizer
    args = sched_params["args"][name]
    if name == "step":
        return [StepLR(optimizer, **args)]
    if name == "multiplicative":
        return [MultiplicativeLR(optimizer, lr_lambda=[lambda1(args["val"])])]
    if name == "lambda":
        return [LambdaLR(optimizer, lr_lambda=[lambda1(args["val"])])]
    raise NotImplementedError("Unimplemented Scheduler!")


def derange_batch(batch_data, modalities, shuffle_time=False, permutation=None):
    """Mismatched-context batch: permute the given modalities along the batch axis (utils.py:85-100)."""
    batch_size = batch_data["p1_face"].size(0)
    if permutation is None:
        permutation = torch.randperm(batch_size)
    dev = batch_data["p1_face"].device
    if dev.type == "cuda" and permutation.device.type == "cpu":
        # (a pageable host-to-device copy synchronises the host with the stream: on the GPU that drains a whole queued training
        # step before the next launch is issued - 0.75 ms of idle per negative step in the step timeline)
        permutation = permutation.pin_memory().to(dev, non_blocking=True)
    else:
        permutation = permutation.to(dev)
    mixed = {}
    for m in ("p1_face", "p2_face", "p1_speech", "p2_speech"):
        if m in modalities:
            mixed[m] = batch_data[m][permutation]
            if shuffle_time:
                t_perm = torch.randperm(batch_data[m].size(1))
                t_perm = t_perm.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else t_perm.to(dev)
                mixed[m] = mixed[m][:, t_perm]
            mixed[m] = mixed[m].contiguous()
        elif batch_data.get(m) is not None:
            mixed[m] = batch_data[m]
    # the reference drops every other key here, so with Conditioning.use_frame_nb its negative-example step dies on
    # batch["frame_nb"] (models.py:541); the counter belongs to the p1 stream and is carried over unchanged
    if batch_data.get("frame_nb") is not None:
        mixed["frame_nb"] = batch_data["frame_nb"]
    return mixed


def get_mismatched_modalities(hparams):
    mods = [m for m in ("p2_face", "p2_speech") if hparams.Conditioning[m]["history"] > 0]
    return mods, ("p2" if len(mods) == 2 else mods[0])


def test_params(hparams):
    for which in (hparams.Train["seq_len"], hparams.Validation["seq_len"]):
        for m in ("p1_face", "p2_face", "p1_speech", "p2_speech"):
            his = hparams.Conditioning[m]["history"] + 1
            assert his < which, f"{his} > {which}"


test_params.__test__ = False  # not a pytest test
