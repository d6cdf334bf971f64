from .models import FeatureEncoder, Glow, SeqGlow  # noqa: F401
from .utils import calc_jerk, get_longest_history  # noqa: F401
