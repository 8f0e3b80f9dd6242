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


# ---- builders: a cut of the Voteview inputs with the matching cut of the reference's own
# stored cube (data/senate_data_90.RData, written by senatedata.R:71), and a cut of the TAPS
# long file -------------------------------------------------------------------------------
import csv  # noqa: E402


def _cols(path, names):
    out = {k: [] for k in names}
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            for k in names:
                out[k].append(row[k])
    return out


ROLL_MAX = 40
mini = {}
for sid in (90, 91):
    v = _cols(os.path.join(REF, "data", "S%d_votes.csv" % sid), ("icpsr", "rollnumber", "cast_code"))
    keep = [i for i, r in enumerate(v["rollnumber"]) if int(float(r)) <= ROLL_MAX]
    for k in v:
        mini["S%d_votes_%s" % (sid, k)] = np.array([int(float(v[k][i])) for i in keep], dtype=np.int32)
    r = _cols(os.path.join(REF, "data", "S%d_rollcalls.csv" % sid), ("rollnumber", "yea_count", "nay_count"))
    keep = [i for i, x in enumerate(r["rollnumber"]) if int(float(x)) <= ROLL_MAX]
    for k in r:
        mini["S%d_rollcalls_%s" % (sid, k)] = np.array([int(float(r[k][i])) for i in keep], dtype=np.int32)
stored = read_rdata(os.path.join(REF, "data", "senate_data_90.RData"))
mini["stored_cube"] = np.asfortranarray(responses_to_int8(stored["data"])[:, :ROLL_MAX, :2])
mini["stored_icpsr"] = np.asarray(stored["unique_icpsr"]).astype(np.int32)
np.savez_compressed(os.path.join(OUT, "senate_mini.npz"), **mini)
print("senate_mini", {k: v.shape for k, v in mini.items()})

t = _cols(os.path.join(REF, "TAPS", "TAPS-econ-data-long.csv"),
          ("WUSTLID", "wave", "ECON2A", "ECON2B", "ECON1A", "ECON1B", "ECON4A", "ECON4B"))
first_ids = []
for w in t["WUSTLID"]:
    if w not in first_ids:
        first_ids.append(w)
    if len(first_ids) == 40:
        break
keep = [i for i, w in enumerate(t["WUSTLID"]) if w in first_ids]
taps = {k: np.array([-1 if t[k][i] in ("NA", "") else int(float(t[k][i])) for i in keep], dtype=np.int32)
        for k in t}
np.savez_compressed(os.path.join(OUT, "taps_mini.npz"), **taps)
print("taps_mini", {k: v.shape for k, v in taps.items()})


# ---- BASELINE config C: the 116th Senate cube with the members' NOMINATE first dimension
# (the score gpirt-senate.R:38-66 initialises from and compares with) -----------------------
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from gpirt_amd.builders import senate_cube  # noqa: E402

c = senate_cube(os.path.join(REF, "data"), [116])
mem = _cols(os.path.join(REF, "data", "S116_members.csv"), ("chamber", "icpsr", "nokken_poole_dim1"))
score = {}
for ch, k, v in zip(mem["chamber"], mem["icpsr"], mem["nokken_poole_dim1"]):
    if ch == "Senate" and v not in ("", "NA"):
        score[int(float(k))] = float(v)
nom = np.array([score.get(int(k), np.nan) for k in c["icpsr"]])
np.savez_compressed(os.path.join(OUT, "senate_116.npz"), data=c["data"], icpsr=c["icpsr"].astype(np.int32),
                    nominate_dim1=nom)
print("senate_116", c["data"].shape, int(np.isfinite(nom).sum()))


# ---- BASELINE config D: the TAPS panel cube (TAPS/load_TAPS.R) ------------------------------
from gpirt_amd.builders import taps_cube  # noqa: E402

t = taps_cube(os.path.join(REF, "TAPS", "TAPS-econ-data-long.csv"))
np.savez_compressed(os.path.join(OUT, "taps_cube.npz"), data=t["data"], ids=t["ids"].astype(np.int64),
                    waves=t["waves"].astype(np.int32))
print("taps_cube", t["data"].shape)
