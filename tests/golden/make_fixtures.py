"""Regenerate tests/golden/*.npz from the reference's data files.

The reference holds no expected-output vectors for the sampler; what it holds is inputs
with their generating ground truth (generate_data.R:229-241).  This script decodes those
workspaces (data only -- nothing is evaluated) into small numpy archives so that the
tests and the GPU box, where /root/reference does not exist, can use them.

    python tests/golden/make_fixtures.py [/root/reference]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from gpirt_amd.rdata import read_rdata, responses_to_int8  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

for stem in ("2PL_C_2_n_100_m_10_h_10_CSTIRF_0_SEED_1",
             "2PL_C_5_n_100_m_10_h_10_CSTIRF_0_SEED_1",
             "Wiener_C_5_n_100_m_10_h_10_CSTIRF_0_SEED_1"):
    d = read_rdata(os.path.join(REF, "data", stem + ".RData"))
    keep = dict(
        y=responses_to_int8(d["data"]),
        y_train=responses_to_int8(d["data_train"]),
        train_idx=np.asfortranarray(d["train_idx"].astype(np.int8)),
        thresholds=d["thresholds"],
        theta=d["theta"],
    )
    for k in ("alpha", "beta", "anchor_xs", "anchor_ys", "SIGMA", "NUM_ANCHOR"):
        if k in d:
            keep[k] = d[k]
    np.savez_compressed(os.path.join(OUT, stem + ".npz"), **keep)
    print(stem, {k: getattr(v, "shape", None) for k, v in keep.items()})
