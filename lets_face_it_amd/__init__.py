"""lets_face_it_amd — MI355X-native conditional-Glow engine behind the lets_face_it Python surface.

`lets_face_it_amd.glow` mirrors `glow_pytorch.glow` of the reference (SeqGlow, LetsFaceItGlow, hparams
handling); the arithmetic runs in liblfi_hip.so (include/lfi.h), sequenced by `lets_face_it_amd.engine`.
"""
__version__ = "0.1.0"
