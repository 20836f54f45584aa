"""Version of the MI355X hot-path build.

``snekmer_version`` on KmerVec objects (reference: snekmer/vectorize.py:231,
snekmer/_version.py) records which implementation produced a ``.kmers`` pickle.
"""
__version__ = "1.3.0+mi355x.r1"
# Reference release whose behaviour the golden fixtures were generated from.
REFERENCE_VERSION = "1.3.0"
