"""Pins the ORACLE (oracle/liboracle.so) against every known answer the
reference's own tests hold for the hot path and against the vectors recorded
from the reference in SURVEY.md Appendix A.  CPU only."""
import pytest

import engines

REF = engines.load_cases("reference_tests.json")
APX = engines.load_cases("survey_appendix_a.json")


@pytest.mark.parametrize("case", REF, ids=[c["id"] for c in REF])
def test_oracle_reference_known_answers(oracle_engine, case):
    assert engines.run_case(oracle_engine, case) == case["expect"], case["src"]


@pytest.mark.parametrize("case", APX, ids=[c["id"] for c in APX])
def test_oracle_appendix_a(oracle_engine, case):
    assert engines.run_case(oracle_engine, case) == case["expect"], case["src"]


def test_oracle_replace_re_rejects_empty_pattern(oracle_engine):
    with pytest.raises(ValueError):
        oracle_engine.replace_re(["a"], "", "x")
