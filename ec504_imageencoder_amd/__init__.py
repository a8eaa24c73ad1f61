"""MI355X-native MPEG-1 I-frame encode path (drop-in for ec504_ImageEncoder's hot path).

Product code lives in csrc/ (HIP kernels + C host driver, built into libencoder.so); this package
is the thin host-side mirror used by tests and bench.py.
"""
from . import _ffi  # noqa: F401
from .encoder import (EncoderError, Mpeg1Encoder, file_prolog, mpeg_encode_procedure,  # noqa: F401
                      set_image_loader)
