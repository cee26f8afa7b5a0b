"""viterbi_spl_amd -- MI355X-native batched Viterbi decoder (hot path of drwangxian/viterbi_spl)."""
__version__ = "0.1.0"
