"""snekmer_amd: MI355X-native implementation of Snekmer's AAR-kmer vectorize + cosine hot path.

Module names mirror the reference package (``snekmer.alphabet``, ``snekmer.vectorize``,
``snekmer.score``, ``snekmer.utils``, ``snekmer.io``) so ``import snekmer_amd as skm`` reads
like ``import snekmer as skm`` in the Snakemake rules.
"""
from . import alphabet, io, score, utils, vectorize  # noqa: F401
from ._version import __version__  # noqa: F401
