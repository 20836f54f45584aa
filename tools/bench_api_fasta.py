#!/usr/bin/env python3
"""tools only: bench.py's `api_vectorize_fasta` measurement alone (kmerize.vectorize_fasta on a synthetic FASTA file of
10 k and 100 k sequences, with and without the compressed .npz write) -> one JSON line."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from snekmer_amd import alphabet
    from snekmer_amd.synth import BASE_SEED

    alphabet.register_alphabet("red6", alphabet.RED6_GROUPS)
    args = argparse.Namespace(length=300, alphabet="red6", k=12)
    print(json.dumps(bench.api_vectorize_fasta(args, BASE_SEED)))


if __name__ == "__main__":
    main()
