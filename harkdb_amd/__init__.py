"""harkdb_amd -- MI355X-native execution layer behind HarkDB's Python surface.

    from harkdb_amd import FutharkContext
    fc = FutharkContext()
    fc.create_table('game_1', 'data.csv')
    fc.sql("select col1, max(col3) from game_1 group by col1")

The compute path is libhark.so (hand-written HIP for gfx950, C ABI in
include/hark.h) loaded through ctypes; importing this package does not touch the
GPU, creating a FutharkContext does and raises if the library or a GPU is missing.
"""
from .context import FutharkContext  # noqa: F401
from .table import Table  # noqa: F401
from .parse import sql_parse  # noqa: F401

__all__ = ["FutharkContext", "Table", "sql_parse"]
