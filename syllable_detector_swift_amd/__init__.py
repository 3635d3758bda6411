"""syllable_detector_swift_amd -- MI355X-native batched syllable detection.

One hot path of gardner-lab/syllable-detector-swift (STFT -> band slice -> sliding window
-> feed-forward net -> threshold) as hand-written HIP kernels for gfx950 behind the C ABI
in include/syldet.h.  Importing this package loads libsyldet.so and fails if it has not
been built: there is no CPU implementation in the product.
"""
from . import _abi
from .config import (InvalidValue, MismatchedLength, MissingValue, NeuralNet, NeuralNetLayer, ParseError,
                     ProcessingFunction, SyllableDetectorConfig, SyllableDetectorError, UnableToOpenPath,
                     createWindow, frequencyIndexRange)
from .bank import PinnedArray, ShardedSyllableDetectorBank, shard_table
from .detector import SyllableDetector
from .resampler import ResamplerLinear, deinterleave

__all__ = ["SyllableDetector", "SyllableDetectorConfig", "NeuralNet", "NeuralNetLayer", "ProcessingFunction",
           "ParseError", "UnableToOpenPath", "MissingValue", "InvalidValue", "MismatchedLength",
           "SyllableDetectorError", "frequencyIndexRange", "createWindow", "ResamplerLinear", "deinterleave",
           "ShardedSyllableDetectorBank", "PinnedArray", "shard_table"]
