"""The I/O contract of code/glow_pytorch/generate_motion_from_model.py (SURVEY.md par. 8f row 3): the 272-d frame layout of
the test-segment tooling -> the model's four streams -> standardise -> SeqGlow.inference -> de-standardise -> the 106-d
FLAME vector (expression 0:100, jaw 100:103, neck 103:106) the untouched renderer takes (mimicry_logger.py:88-108,
visualize/render_server.py:31-62). The shipped script is stale (it imports modules that do not exist and reads
`model.face_means`, which nothing defines: SURVEY.md par. 2 #11), so the means / stds are explicit arguments here: they
are the `/means` and `/stds` groups of the dataset file (combine_features.py:258-261)."""
import torch

from .glow.utils import get_longest_history


def get_face_indicies(exp_dim, jaw_dim, neck_dim, offset=0):
    """misc/utils.py:38-43."""
    return (list(range(offset, offset + exp_dim)) + list(range(100 + offset, 100 + offset + jaw_dim))
            + list(range(103 + offset, 103 + offset + neck_dim)))


def dictify_frames(frames, data_hparams):
    """(T, 272) frames -> {"p1_face", "p1_speech", "p2_face", "p2_speech"} (generate_motion_from_model.py:73-87)."""
    e, j, n, s = (data_hparams[k] for k in ("expression_dim", "jaw_dim", "neck_dim", "speech_dim"))
    left = torch.tensor(get_face_indicies(e, j, n), device=frames.device)
    right = torch.tensor(get_face_indicies(e, j, n, offset=136), device=frames.device)
    return {"p1_face": frames.index_select(1, left), "p1_speech": frames[:, 106:106 + s],
            "p2_face": frames.index_select(1, right), "p2_speech": frames[:, 242:242 + s]}


def expand_face_dim(seq, data_hparams):
    """(B, T, e+j+n) model output -> (B, T, 106) FLAME parameters (generate_motion_from_model.py:39-51)."""
    e, j, n = (data_hparams[k] for k in ("expression_dim", "jaw_dim", "neck_dim"))
    out = torch.zeros((seq.size(0), seq.size(1), 106), dtype=seq.dtype, device=seq.device)
    out[:, :, :e] = seq[:, :, :e]
    out[:, :, 100:100 + j] = seq[:, :, e:e + j]
    out[:, :, 103:103 + n] = seq[:, :, e + j:e + j + n]
    return out


def get_data(data, model, stats, use_zero_pose=True):
    """Standardise the four streams and cut the p1_face seed (generate_motion_from_model.py:16-36).
    stats: {"face_means", "face_stds", "speech_means", "speech_stds"} tensors."""
    def standardize(x, m, s):
        return ((x - m) / s).unsqueeze(0)

    seed = data["p1_face"][:get_longest_history(model.hparams.Conditioning)]
    p1_face = torch.zeros_like(seed).unsqueeze(0) if use_zero_pose else standardize(seed, stats["face_means"], stats["face_stds"])
    return {"p1_face": p1_face.contiguous(),
            "p2_face": standardize(data["p2_face"], stats["face_means"], stats["face_stds"]).contiguous(),
            "p1_speech": standardize(data["p1_speech"], stats["speech_means"], stats["speech_stds"]).contiguous(),
            "p2_speech": standardize(data["p2_speech"], stats["speech_means"], stats["speech_stds"]).contiguous()}


def generate_motion(frames, model, stats, eps=1, noise=None):
    """frames: (T, 272) on the model's device -> (1, T - start, 106) de-standardised FLAME parameters
    (generate_motion_from_model.py:54-70; `model` is a loaded LetsFaceItGlow instead of a checkpoint path)."""
    model.hparams.Infer["eps"] = eps
    model.eval()
    seq_len = frames.size(0)
    cond_data = get_data(dictify_frames(frames.float(), model.hparams.Data), model, stats, use_zero_pose=True)
    predicted = model.seq_glow.inference(seq_len, data=cond_data, noise=noise)
    return expand_face_dim(predicted * stats["face_stds"] + stats["face_means"], model.hparams.Data)
