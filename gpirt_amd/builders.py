"""Response cubes from the reference's raw data files, without R.

`senate_cube`  Voteview roll-call CSVs -> [members, roll calls, congresses]   (senatedata.R:13-71)
`taps_cube`    TAPS long-format survey CSV -> [panelists, questions, waves]   (TAPS/load_TAPS.R:8-46)

Both return int8 cubes in the library's convention (categories 1..C, 0 = missing) plus the
row / column keys.  The R scripts fill their cubes with a triple loop of data-frame lookups;
here the same cells are placed with one vectorised scatter per file.
"""
import csv
import os

import numpy as np

# members the reference leaves out (senatedata.R:20-27): BARKLEY (40106), RUSSELL (8138)
SENATE_DROP_ICPSR = (40106, 8138)


def _read_csv_columns(path, columns, convert):
    """Columns of a CSV file as numpy arrays (quoted fields with commas handled by csv)."""
    out = {c: [] for c in columns}
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            for c in columns:
                out[c].append(row[c])
    return {c: np.array([convert(v) for v in out[c]]) for c in columns}


def _to_int(v):
    return int(float(v))          # some exports write integer keys as "14226.0"


def _int_or_nan(v):
    return float(v) if v not in ("", "NA") else np.nan


def _ordered_unique(a):
    """Values of `a` in order of first appearance (R's unique())."""
    _, first = np.unique(a, return_index=True)
    return a[np.sort(first)]


def senate_cube(data_dir, session_ids, drop_icpsr=SENATE_DROP_ICPSR, exclude_rollcalls=None,
                width=None):
    """Roll-call cube of the U.S. Senate (senatedata.R:13-71).

    data_dir          directory holding ``S<congress>_votes.csv`` / ``S<congress>_rollcalls.csv``
    session_ids       congress numbers, e.g. range(90, 100); one period each
    drop_icpsr        members left out (:20-27)
    exclude_rollcalls optional {congress: iterable of roll numbers} to leave out -- the
                      reference drops the abortion votes listed in ``SenatePeriods.dta`` (:40-41),
                      a file that is not distributed with it
    width             number of roll-call columns (the reference fixes 1202, :30); default: the
                      largest number of kept roll calls in any congress

    Returns ``dict(data=int8 [n, width, H], icpsr=int [n], rollnumbers=[H lists])``: yea
    (cast codes 1-3) = 2, nay (4-6) = 1, everything else missing (:54-60).  Members are in
    order of first appearance over the congresses (:13-18), roll calls in file order after
    removing unanimous ones (:37).
    """
    session_ids = list(session_ids)
    votes = {}
    members = []
    for sid in session_ids:
        v = _read_csv_columns(os.path.join(data_dir, "S%d_votes.csv" % sid),
                              ("icpsr", "rollnumber", "cast_code"), _to_int)
        votes[sid] = v
        members.append(_ordered_unique(v["icpsr"]))
    icpsr = _ordered_unique(np.concatenate(members))
    icpsr = icpsr[~np.isin(icpsr, np.array(list(drop_icpsr), dtype=icpsr.dtype))]
    row_of = {int(k): i for i, k in enumerate(icpsr)}

    kept = []
    for sid in session_ids:
        rc = _read_csv_columns(os.path.join(data_dir, "S%d_rollcalls.csv" % sid),
                               ("rollnumber", "yea_count", "nay_count"), _to_int)
        ok = (rc["yea_count"] != 0) & (rc["nay_count"] != 0)
        rn = _ordered_unique(rc["rollnumber"][ok])
        if exclude_rollcalls and sid in exclude_rollcalls:
            rn = rn[~np.isin(rn, np.array(list(exclude_rollcalls[sid])))]
        kept.append(rn)
    if width is None:
        width = max(len(r) for r in kept)
    data = np.zeros((len(icpsr), width, len(session_ids)), dtype=np.int8, order="F")
    for h, sid in enumerate(session_ids):
        rn = kept[h][:width]
        col_of = {int(r): j for j, r in enumerate(rn)}
        v = votes[sid]
        rows = np.array([row_of.get(int(k), -1) for k in v["icpsr"]])
        cols = np.array([col_of.get(int(r), -1) for r in v["rollnumber"]])
        code = v["cast_code"]
        val = np.where((code >= 1) & (code <= 3), 2, np.where((code >= 4) & (code <= 6), 1, 0)).astype(np.int8)
        use = (rows >= 0) & (cols >= 0) & (val > 0)
        data[rows[use], cols[use], h] = val[use]
    return dict(data=data, icpsr=icpsr, rollnumbers=[list(map(int, r)) for r in kept])


TAPS_QUESTIONS = ("ECON2A", "ECON2B", "ECON1A", "ECON1B", "ECON4A", "ECON4B")


def taps_cube(path, min_wave=20, min_answers=165, questions=TAPS_QUESTIONS):
    """Panel cube of the TAPS economy items (TAPS/load_TAPS.R:8-46).

    Keeps waves >= `min_wave` (:10), panelists with at least `min_answers` non-missing answers
    over the kept waves and questions (:19-25; 168 in the reference's TEST_YEAR runs, :27-29).
    Returns ``dict(data=int8 [n, m, H], ids, waves)`` with answers 1..5 and 0 = missing.

    The reference initialises theta from a graded-response fit of R's ``ltm`` package
    (:53-58), which is not available here; `standardised_row_scores` is the substitute.
    """
    cols = _read_csv_columns(path, ("WUSTLID", "wave") + tuple(questions), _int_or_nan)
    wid, wave = cols["WUSTLID"].astype(np.int64), cols["wave"].astype(np.int64)
    ans = np.stack([cols[q] for q in questions], axis=1)            # [rows, m], NaN = NA
    keep = wave >= min_wave
    wid, wave, ans = wid[keep], wave[keep], ans[keep]
    ids = _ordered_unique(wid)
    waves = _ordered_unique(wave)
    answered = np.array([np.isfinite(ans[wid == i]).sum() for i in ids])
    ids = ids[answered >= min_answers]
    row_of = {int(k): i for i, k in enumerate(ids)}
    per_of = {int(w): h for h, w in enumerate(waves)}
    data = np.zeros((len(ids), len(questions), len(waves)), dtype=np.int8, order="F")
    for r in range(wid.size):
        i = row_of.get(int(wid[r]))
        if i is None:
            continue
        a = ans[r]
        data[i, :, per_of[int(wave[r])]] = np.where(np.isfinite(a), a, 0).astype(np.int8)
    return dict(data=data, ids=ids, waves=waves)


def standardised_row_scores(data):
    """theta_init [n, H] when no IRT fit is at hand: each respondent's mean answer per period,
    standardised over respondents (periods without answers get 0)."""
    d = np.asarray(data, dtype=np.float64)
    obs = d > 0
    cnt = obs.sum(axis=1)
    mean = np.where(cnt > 0, (d * obs).sum(axis=1) / np.maximum(cnt, 1), np.nan)      # [n, H]
    has = np.isfinite(mean)
    k = np.maximum(has.sum(axis=0, keepdims=True), 1)
    mu = np.where(has, mean, 0.0).sum(axis=0, keepdims=True) / k
    sd = np.sqrt(np.where(has, (mean - mu) ** 2, 0.0).sum(axis=0, keepdims=True) / k)
    return np.where(has, (mean - mu) / np.where(sd > 0, sd, 1.0), 0.0)
