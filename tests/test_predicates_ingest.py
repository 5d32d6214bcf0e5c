"""Host logic fixed in round 2: WHERE/HAVING literals that the column's dtype cannot hold, and CSV ingest of
mixed int/float files.  CPU only."""
import itertools

import numpy as np
import pytest

from harkdb_amd.engine import normalise_predicate

_NP = {">": np.greater, ">=": np.greater_equal, "<": np.less, "<=": np.less_equal, "=": np.equal, "!=": np.not_equal}


@pytest.mark.parametrize("dt", [np.int32, np.uint32, np.int64])
def test_normalised_predicate_selects_what_exact_arithmetic_selects(dt):
    info = np.iinfo(dt)
    col = np.array([info.min, info.min + 1, -3 if info.min < 0 else 3, 0, 1, 2, 3, 1000, info.max - 1, info.max], dtype=dt)
    literals = [2.5, -2.5, 2.0, 0.0, -0.0, 3, 2, 1e30, -1e30, float("inf"), float("-inf"), float("nan"), 3000000000, -3000000000,
                2**63, -2**63 - 1, int(info.max), int(info.min), int(info.max) + 1, int(info.min) - 1, 0.999999, np.float32(2.5), np.int64(7), True]
    for cmp, lit in itertools.product(_NP, literals):
        c2, const = normalise_predicate(dt, cmp, lit)
        assert const.dtype == np.dtype(dt) and const.shape == (1,)
        got = _NP[c2](col, const[0])
        # exact semantics: compare the integers with the literal as Python numbers (arbitrary precision ints, exact floats)
        pyl = lit.item() if isinstance(lit, np.generic) else lit
        exp = np.array([{">": x > pyl, ">=": x >= pyl, "<": x < pyl, "<=": x <= pyl, "=": x == pyl, "!=": x != pyl}[cmp] for x in col.tolist()])
        assert np.array_equal(got, exp), (dt, cmp, lit, c2, const)


def test_float_columns_keep_the_literal():
    c, const = normalise_predicate(np.float32, "<", 2.5)
    assert c == "<" and const.dtype == np.float32 and const[0] == np.float32(2.5)
    assert normalise_predicate(np.float32, "==", 1)[0] == "="
    with pytest.raises(KeyError):
        normalise_predicate(np.int32, "~", 1)


def test_mixed_csv_keeps_integer_columns_exact(tmp_path):
    """ids above 2^24 next to a float column: `.values` of the frame is float64 and the ids would be rounded to f32."""
    from harkdb_amd.table import Table
    f = tmp_path / "mixed.csv"
    f.write_text("id, x, big\n16777217, 0.5, 1099511627776\n16777219, 1.25, -5\n3, 2.0, 7\n")
    t = Table("m", str(f))
    assert t.get_schema() == ["id", "x", "big"]
    cols = t.host_columns()
    assert [c.dtype for c in cols] == [np.int32, np.float32, np.int64]
    assert cols[0].tolist() == [16777217, 16777219, 3]
    assert cols[2].tolist() == [1099511627776, -5, 7]
    assert t.get_data().shape == (3, 3)


def test_homogeneous_csv_unchanged():
    from conftest import GOLDEN, data_csv
    from harkdb_amd.table import Table
    t = Table("game_1", f"{GOLDEN}/data.csv")
    assert np.array_equal(t.get_data(), data_csv())
    assert all(c.dtype == np.int32 for c in t.host_columns())
