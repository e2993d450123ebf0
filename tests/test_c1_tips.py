"""Config C1: the reference README's tips.csv pipeline (README.md:17-39) -- split(','),
seven literal day-token replaces, category -- on the synthesised 244-row tips-like CSV.
CPU: the oracle against pandas.Series.str (the mirror BASELINE.json names); GPU: the product
against the oracle."""
import pytest

import c1_tips
import engines


def test_c1_oracle_matches_pandas():
    host = c1_tips.lines()
    assert len(host) == 244
    got = c1_tips.pipeline(engines.OracleEngine(), host)
    exp = c1_tips.pandas_pipeline(host)
    assert len(got["columns"]) == 7
    assert got == exp
    assert got["keys"] == ["Fri", "Sat", "Sun", "Thur"]  # README.md:43
    assert set(got["day_encoded"]) == {"0", "4", "5", "6"}


def test_c1_row_emulation_matches_oracle():
    host = c1_tips.lines(seed=7)
    emu, orc = engines.EmuEngine(), engines.OracleEngine()
    cols = emu.split(host, ",", -1)
    assert cols == orc.split(host, ",", -1)
    day = cols[4]
    for idx, d in enumerate(c1_tips.DAYS):
        day = emu.replace(day, d, str(idx), -1)
    assert day == c1_tips.pipeline(orc, host)["day_encoded"]


@pytest.mark.gpu
def test_gpu_c1_tips_pipeline(gpu_engine, oracle_engine):
    for seed in (20240607, 1):
        host = c1_tips.lines(seed)
        assert c1_tips.pipeline(gpu_engine, host) == c1_tips.pipeline(oracle_engine, host)
    # with a header-less empty line and a null row in the middle (to_device(None) -> null)
    host = c1_tips.lines(3)
    host[10] = None
    host[11] = ""
    assert c1_tips.pipeline(gpu_engine, host) == c1_tips.pipeline(oracle_engine, host)
