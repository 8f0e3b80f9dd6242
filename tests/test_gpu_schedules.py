"""The multi-stream schedules of the dense route against the CPU oracle DIRECTLY.

From 4096 inputs on the factorisation may run with rows of nu = L Z (elliptical_slice.m:64,
sample_fs.m:2) beside it on a second stream, with the earlier panels' share of the last rows
added there too, and with the bulk of each trailing update on that stream (look-ahead).  The
full-size tests hold those schedules to the plain one bit for bit; here one sweep of each forced
combination (main.m:46-87) is compared with the oracle's sweep at a size the oracle still
finishes in well under a minute: same slice decisions (equal evaluation counters), theta / beta
/ thresholds to 1e-9, f / f* to the accuracy of the solves."""
import os

import numpy as np
import pytest

from gpirt_amd import _abi
from test_gpu_parity import assert_states_agree

pytestmark = pytest.mark.gpu

ENV = ("GPIRT_NU_OVERLAP", "GPIRT_LOOKAHEAD", "GPIRT_NU_TAIL")


def _device_sweeps(monkeypatch, cfg, d, sweeps, overlap, ahead, tail):
    from gpirt_amd.sampler import Sampler
    for name, v in zip(ENV, (overlap, ahead, tail)):
        if v is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, str(v))
    states = []
    with Sampler(cfg, d["y"], theta_init=d["theta"]) as s:
        for _ in range(sweeps):
            s.sweep(1)
            states.append(s.state())
        return states, s.diagnostics(), s.nu_rows_beside_factor(), s.factor_lookahead(), s.nu_tail_panels()


def _oracle_sweeps(cfg, d, sweeps):
    from oracle import oracle
    states = []
    c = oracle.Sampler(cfg, d["y"], theta_init=d["theta"])
    for _ in range(sweeps):
        c.sweep(1)
        states.append(c.state())
    dg = c.diagnostics()
    c.close()
    return states, dg


def test_rows_and_panels_of_nu_beside_the_factorisation_match_the_oracle(monkeypatch):
    """4200 respondents x 4000 items, C = 5: 33 block columns, 9 super-panels, 32 column tiles per
    row panel (the single-launch form of nu).  Rows of six super-panels beside the factorisation
    with the first four panels' share of the other rows, without and with look-ahead."""
    from gpirt_amd.synth import generate
    n, m, C = 4200, 4000, 5
    d = generate(n, m, C, seed=7)
    cfg = _abi.default_config(n=n, m=m, H=1, C=C, seed=12)
    want, dg_c = _oracle_sweeps(cfg, d, 1)
    for ahead in (0, 1):
        got, dg, rows, la, tail = _device_sweeps(monkeypatch, cfg, d, 1, 6, ahead, 4)
        assert rows == 6 * 512 and bool(la) == bool(ahead) and tail == 4, (rows, la, tail)
        assert_states_agree(got[0], want[0], ("look-ahead", ahead))
        assert dg == dg_c


def test_slabs_of_nu_beside_the_factorisation_match_the_oracle(monkeypatch):
    """4400 respondents x 160 items (a rank's share): nu in 8 slabs of K side by side, the rows of
    seven of the nine super-panels computed that way beside the factorisation, look-ahead forced;
    two sweeps."""
    from gpirt_amd.synth import generate
    n, m, C = 4400, 160, 5
    d = generate(n, m, C, seed=8)
    cfg = _abi.default_config(n=n, m=m, H=1, C=C, seed=13)
    want, dg_c = _oracle_sweeps(cfg, d, 2)
    got, dg, rows, la, _ = _device_sweeps(monkeypatch, cfg, d, 2, 7, 1, None)
    assert rows == 7 * 512 and la
    for sweep in range(2):
        assert_states_agree(got[sweep], want[sweep], sweep)
    assert dg == dg_c
