"""Regenerates the fixtures in tests/golden/.  Run in the BUILD container only
(it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_fixtures.py

Provenance of each file, stated honestly:
  cats_corpus.npz      DATA from the reference: integer encoding (type id = order of first
                       appearance) of src/main/resources/datasets/cats.txt -- 23 documents,
                       7 788 tokens, 303 types.  No reference source text is copied.
  kat_vectors.json     Published known answers, NOT produced by this repo's code:
                       Philox4x32-10 (Random123 kat_vectors), java.util.Random (JDK javadoc
                       algorithm; widely published values for seeds 0 and 42), and the
                       thetaEstimate / z-bar inputs of the reference's own
                       src/test/java/cc/mallet/topics/ModifiedSimpleLDATest.java:10-26.
  cats_ggs_golden.npz  RESTATEMENT-DERIVED: outputs of oracle/ggs_oracle.c on cats (cfg of
                       plda-cats-test.cfg:18-25: alpha=5, beta=7, seed=2019; K=3 and K=20).
                       The reference's GGS path has an unseedable RNG and no JVM exists here, so
                       these pin the oracle against ITSELF across rebuilds/refactors and pin the
                       HIP path against the oracle; they are not Java outputs ("parity unpinned"
                       against a JVM run -- see DESIGN.md).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from ldagroupedgibbssampler_amd.corpus import load_tsv_corpus  # noqa: E402
from oracle import oracle as O  # noqa: E402

REF_CATS = "/root/reference/src/main/resources/datasets/cats.txt"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    c = load_tsv_corpus(REF_CATS)
    assert (c.num_docs, c.num_tokens, c.num_types) == (23, 7788, 303)
    np.savez_compressed(os.path.join(HERE, "cats_corpus.npz"), doc_ptr=c.doc_ptr, tokens=c.tokens,
                        num_types=np.int64(c.num_types))

    kat = {
        "philox4x32_10": [
            {"ctr": [0, 0, 0, 0], "key": [0, 0], "out": [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]},
            {"ctr": [0xffffffff] * 4, "key": [0xffffffff] * 2, "out": [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]},
            {"ctr": [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], "key": [0xa4093822, 0x299f31d0],
             "out": [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]},
        ],
        "java_util_random": {
            "nextInt_seed42": [-1170105035, 234785527],
            "nextInt_seed0": [-1155484576, -723955400],
            "nextInt10_seed42": [0, 3, 8, 4, 0],
            "nextDouble_seed42": [0.7275636800328681],
        },
        "modified_simple_lda_test": {
            "numTopics": 5, "alpha": 0.01, "alphas": [0.01, 5.0, 0.04, 0.1, 0.000001],
            "docLength": 10, "oneDocTopics": [1, 2, 0, 1, 0, 1, 2, 0, 1, 0],
        },
    }
    with open(os.path.join(HERE, "kat_vectors.json"), "w") as f:
        json.dump(kat, f, indent=1)

    out = {}
    for K in (3, 20):
        s = O.OracleSampler(K, c.num_types, 5.0, 7.0, seed=2019)
        s.set_corpus(c.doc_ptr, c.tokens)
        s.init_z_java_lcg(2019)
        out["K%d_z0" % K] = s.get_z()
        s.init_phi()
        out["K%d_phi0_sha256" % K] = np.frombuffer(bytes.fromhex(sha(s.get_phi())), np.uint8)
        s.sweep(3)
        out["K%d_z3" % K] = s.get_z()
        out["K%d_nk3" % K] = s.get_topic_totals()
        out["K%d_nwk3" % K] = s.get_type_topic_counts()
        out["K%d_phi3_sha256" % K] = np.frombuffer(bytes.fromhex(sha(s.get_phi())), np.uint8)
        out["K%d_theta3_sha256" % K] = np.frombuffer(bytes.fromhex(sha(s.get_theta())), np.uint8)
        out["K%d_phi3_row0_head" % K] = s.get_phi()[0, :8].copy()
        out["K%d_theta3_doc0" % K] = s.get_theta()[0].copy()
    np.savez_compressed(os.path.join(HERE, "cats_ggs_golden.npz"), **out)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
