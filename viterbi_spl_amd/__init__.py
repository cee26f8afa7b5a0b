"""viterbi_spl_amd -- MI355X-native batched Viterbi decoder (hot path of drwangxian/viterbi_spl).

Importing the package never touches the GPU; the HIP library is loaded on first use and there is
no CPU fallback (see ``_lib.load``).
"""
__version__ = "0.2.0"

from .decoder import ViterbiDecoder, decode, get_decoder  # noqa: F401
from .reference_api import (  # noqa: F401
    RecordingAccumulator, ScaledSoftMaxViterbi, SoftMaxViterbi, Viterbi, tf_viterbi_librosa_fn, viterbi_librosa_c_fn,
    viterbi_librosa_fn, viterbi_numba_core, viterbi_numba_fn, viterbi_tf_fn,
)
