"""
kaldi_tflite_amd — MI355X-native drop-in for the wav -> x-vector hot path of
shahruk10/kaldi-tflite.  `import kaldi_tflite_amd as ktf` gives the same surface the
reference exposes as `import kaldi_tflite as ktf` for that path: ktf.layers, ktf.models,
ktf.io, ktf.kaldi_numpy.
"""

from . import io, kaldi_numpy, layers, models, parallel  # noqa: F401
from ._lib import KtfBackendError  # noqa: F401

__version__ = "0.1.0"
