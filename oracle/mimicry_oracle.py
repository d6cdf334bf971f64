"""CPU restatement (numpy) of the reference's window dataset and validation metrics — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(lets_face_it_amd/) never does. It follows, line by line:

  MimicryDataset.__init__ / __getitem__   /root/reference/code/glow_pytorch/mimicry_data_module.py:33-78
  the HDF5 layout it reads                 /root/reference/code/feature_extraction/combine_features.py:246-265
  calc_jerk                                /root/reference/code/glow_pytorch/glow/utils.py:53-58
  expand_face_dim / dictify_frames         /root/reference/code/glow_pytorch/generate_motion_from_model.py:39-51,73-87
  get_face_indicies                        /root/reference/code/misc/utils.py:38-43

Pinning: PINNED against the reference itself. tests/golden/make_golden.py (`mimicry`) imports the reference's
mimicry_data_module.py / glow/utils.py / misc/utils.py / generate_motion_from_model.py in the build container (h5py.File is
replaced by a dict-backed stand-in: the dataset only does `File(name)[split][kind][key][who][rows]`), runs them on a small
seeded corpus, asserts this restatement equal (bit-exact for the copies and index lists, 1e-6 for the jerk mean) and writes
tests/golden/mimicry.npz; tests/test_data_module.py re-checks the restatement against that file everywhere. A `store` below
is the HDF5 tree as nested dicts: store[split][kind][bin_key][who] -> (len, dim) float array.
"""
import numpy as np

KINDS = ("flame_expression", "flame_jaw", "flame_neck", "mfcc", "prosody")


def window_index(store, data_type, seq_len):
    """mimicry_data_module.py:33-41: every stride-1 window of seq_len frames of every bin with at least seq_len frames, in
    bin order then start order (before the reference's random.sample shuffle, :43). -> list of (bin_key, start)."""
    out = []
    for key, chunk in store[data_type]["prosody"].items():
        n = len(chunk["agent"])
        if n >= seq_len:
            for start in range(n - seq_len + 1):      # torch.arange(n).unfold(0, seq_len, 1)
                out.append((key, start))
    return out


def get_item(store, data_type, key, start, seq_len, expression_dim, histories):
    """mimicry_data_module.py:45-78. histories: {"p1_speech": h, "p2_speech": h, "p2_face": h} (0 = modality absent)."""
    rows = slice(start, start + seq_len)

    def item(kind, who):
        return np.asarray(store[data_type][kind][key][who])[rows]

    def face(who):  # (:52-60) expression[:, :expression_dim] | jaw | neck
        return np.concatenate([item("flame_expression", who)[:, :expression_dim], item("flame_jaw", who),
                               item("flame_neck", who)], axis=1).astype(np.float32)

    def speech(who):  # (:62-65) mfcc | prosody
        return np.concatenate([item("mfcc", who), item("prosody", who)], axis=1).astype(np.float32)

    d = {"p1_face": face("agent")}
    if histories.get("p1_speech"):
        d["p1_speech"] = speech("agent")
    if histories.get("p2_speech"):
        d["p2_speech"] = speech("interlocutor")
    if histories.get("p2_face"):
        d["p2_face"] = face("interlocutor")
    return d


def calc_jerk(x):
    """glow/utils.py:53-58 on a (B, T, C) array: mean absolute third difference along time (fp32 differences)."""
    x = np.asarray(x, dtype=np.float32)
    d1 = x[:, 1:] - x[:, :-1]
    d2 = d1[:, 1:] - d1[:, :-1]
    d3 = d2[:, 1:] - d2[:, :-1]
    return np.abs(d3).astype(np.float64).mean()


def get_face_indicies(exp_dim, jaw_dim, neck_dim, offset=0):
    """misc/utils.py:38-43."""
    return (list(range(offset, offset + exp_dim)) + list(range(100 + offset, 100 + offset + jaw_dim))
            + list(range(103 + offset, 103 + offset + neck_dim)))


def dictify_frames(frames, data_hparams):
    """generate_motion_from_model.py:73-87: the 272-d frame layout -> the model's four streams."""
    e, j, n, s = (data_hparams[k] for k in ("expression_dim", "jaw_dim", "neck_dim", "speech_dim"))
    return {"p1_face": frames[:, get_face_indicies(e, j, n)], "p1_speech": frames[:, 106:106 + s],
            "p2_face": frames[:, get_face_indicies(e, j, n, offset=136)], "p2_speech": frames[:, 242:242 + s]}


def expand_face_dim(seq, data_hparams):
    """generate_motion_from_model.py:39-51: (B, T, e+j+n) -> the 106-d FLAME vector (expr 0:100, jaw 100:103, neck 103:106)."""
    e, j, n = (data_hparams[k] for k in ("expression_dim", "jaw_dim", "neck_dim"))
    out = np.zeros((seq.shape[0], seq.shape[1], 106), dtype=np.float32)
    out[:, :, :e] = seq[:, :, :e]
    out[:, :, 100:100 + j] = seq[:, :, e:e + j]
    out[:, :, 103:103 + n] = seq[:, :, e + j:e + j + n]
    return out
