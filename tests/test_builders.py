"""Data builders against the reference's own stored cube (data/senate_data_90.RData, written
by senatedata.R:71) and against a literal transcription of TAPS/load_TAPS.R:8-46."""
import csv
import os

import numpy as np
import pytest

from gpirt_amd.builders import senate_cube, standardised_row_scores, taps_cube

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _write_csv(path, cols):
    names = list(cols)
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(names)
        for i in range(len(cols[names[0]])):
            w.writerow([cols[k][i] for k in names])


def test_senate_cube_matches_the_cut_of_the_references_stored_cube(tmp_path):
    d = np.load(os.path.join(HERE, "golden", "senate_mini.npz"))
    for sid in (90, 91):
        _write_csv(tmp_path / ("S%d_votes.csv" % sid),
                   {k: d["S%d_votes_%s" % (sid, k)] for k in ("icpsr", "rollnumber", "cast_code")})
        _write_csv(tmp_path / ("S%d_rollcalls.csv" % sid),
                   {k: d["S%d_rollcalls_%s" % (sid, k)] for k in ("rollnumber", "yea_count", "nay_count")})
    c = senate_cube(str(tmp_path), (90, 91))
    full_row = {int(k): i for i, k in enumerate(d["stored_icpsr"])}
    assert set(c["data"].ravel()) <= {0, 1, 2}
    for h in range(2):
        k = len(c["rollnumbers"][h])
        assert k > 30                                   # a few of the 40 roll calls are unanimous
        for i, icpsr in enumerate(c["icpsr"]):
            assert np.array_equal(c["data"][i, :k, h], d["stored_cube"][full_row[int(icpsr)], :k, h])
    # members of the stored cube that are not in the cut never voted in it
    absent = [full_row[int(k)] for k in d["stored_icpsr"] if int(k) not in set(map(int, c["icpsr"]))]
    kmin = min(len(r) for r in c["rollnumbers"])
    assert not d["stored_cube"][absent, :kmin, :].any()


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "data", "senate_data_90.RData")),
                    reason="needs the reference's data directory")
def test_senate_cube_equals_the_references_stored_cube():
    from gpirt_amd.rdata import read_rdata, responses_to_int8
    stored = read_rdata(os.path.join(REF, "data", "senate_data_90.RData"))
    c = senate_cube(os.path.join(REF, "data"), range(90, 100), width=1202)
    assert np.array_equal(c["icpsr"], np.asarray(stored["unique_icpsr"]).astype(np.int64))
    assert np.array_equal(c["data"], responses_to_int8(stored["data"]))


def test_taps_cube_follows_the_r_loops(tmp_path):
    d = np.load(os.path.join(HERE, "golden", "taps_mini.npz"))
    qs = ("ECON2A", "ECON2B", "ECON1A", "ECON1B", "ECON4A", "ECON4B")
    cols = {k: ["NA" if v < 0 else int(v) for v in d[k]] for k in ("WUSTLID", "wave") + qs}
    _write_csv(tmp_path / "taps.csv", cols)
    t = taps_cube(str(tmp_path / "taps.csv"), min_wave=20, min_answers=100)
    # TAPS/load_TAPS.R:8-46, literally
    rows = [i for i in range(len(d["wave"])) if d["wave"][i] >= 20]
    ids, waves = [], []
    for i in rows:
        if d["WUSTLID"][i] not in ids:
            ids.append(int(d["WUSTLID"][i]))
        if d["wave"][i] not in waves:
            waves.append(int(d["wave"][i]))
    freq = {k: 0 for k in ids}
    for i in rows:
        freq[int(d["WUSTLID"][i])] += sum(int(d[q][i] >= 0) for q in qs)
    ids = [k for k in ids if freq[k] >= 100]
    assert list(t["ids"]) == ids and list(t["waves"]) == waves and len(ids) > 5
    first_row = {}
    for i in rows:
        first_row.setdefault((int(d["wave"][i]), int(d["WUSTLID"][i])), i)
    for h, w in enumerate(waves):
        for a, k in enumerate(ids):
            r = first_row.get((w, k))
            for j, q in enumerate(qs):
                want = 0 if r is None or d[q][r] < 0 else int(d[q][r])
                assert t["data"][a, j, h] == want
    z = standardised_row_scores(t["data"])
    assert z.shape == (len(ids), len(waves)) and np.isfinite(z).all()
    assert np.abs(z.mean(axis=0)).max() < 1e-12
