"""Table ingest: CSV / TXT / DataFrame / ndarray -> (host columns, headers).

Mirrors the reference's table.py:8-80 (`Table`, `load_table` and the three
loaders) with two differences the reference's own docstring asks for
(table.py:23 "datatype i64", operators i32/u32):
  * columns are typed individually (int32 when the values fit, else int64;
    floating-point columns become float32), so a DataFrame with mixed columns
    loads;
  * default ndarray headers count COLUMNS (`shape[1]`); the reference counts
    rows (table.py:14, a bug: the header list then has the wrong length).
"""
import numpy as np
import pandas as pd


def load_df(df):
    # table.py:8-10
    return df.to_numpy(), list(df)


def load_np(nparray, col_names=None):
    # table.py:12-16 (with the shape[1] fix described above)
    if col_names is None:
        ncols = nparray.shape[1] if nparray.ndim == 2 else 1
        return nparray, ["col" + str(i + 1) for i in range(ncols)]
    return nparray, col_names


def load_file(file_name, col_names=None):
    # table.py:18-40
    if file_name[-3:] == "csv":
        table = pd.read_csv(file_name, skipinitialspace=True)
        headers = [str(h).strip() for h in table.columns.tolist()]
        table.columns = headers
        return (table, headers)          # the frame itself: `.values` would upcast a mixed int/float file to float64
    if file_name[-3:] == "txt":
        table = np.loadtxt(file_name, ndmin=2)
        headers = ["c" + str(i + 1) for i in range(table.shape[1])] if col_names is None else col_names
        return (table, headers)
    raise Exception("We do not support loading this file type")


def load_table(table_name, table):
    # table.py:42-50
    if isinstance(table, pd.DataFrame):
        return load_df(table)
    if isinstance(table, np.ndarray):
        return load_np(table)
    if isinstance(table, str):
        return load_file(table)
    raise Exception("Table is not in a file, numpy array or dataframe")


_warned_narrowing = False


def column_dtype(col):
    """Device dtype for one host column."""
    col = np.asarray(col)
    if col.dtype.kind == "f":
        if col.dtype.itemsize > 4 and col.size and not _warned_narrowing:
            # float64 in, float32 on the device (the fused kernels' value type): integers above 2^24 and 9+ significant digits
            # do not survive.  Said once per process; INTEGRATION.md "dtypes" has the whole table.
            if not np.array_equal(col.astype(np.float32).astype(col.dtype), col, equal_nan=True):
                import warnings
                warnings.warn(f"harkdb_amd: a {col.dtype} column is stored as float32 on the device and loses precision "
                              "(cast it yourself, or scale it to integers, to choose how)", stacklevel=3)
                globals()["_warned_narrowing"] = True
        return np.float32
    if col.dtype.kind in "iub":
        if col.dtype == np.uint32:
            return np.uint32
        if col.size == 0 or (col.min() >= -2**31 and col.max() < 2**31):
            return np.int32
        return np.int64
    raise Exception(f"unsupported column dtype {col.dtype}")


class Table:
    """table.py:52-80: a schema and the data."""

    def __init__(self, table_name, file_name):
        self._table_name = table_name
        table, headers = load_table(table_name, file_name)
        # a DataFrame (given, or read from a CSV) keeps its per-column dtypes: df.to_numpy() upcasts a mixed
        # int/float frame to float64, which would turn integer keys above 2^24 into wrong f32 values
        frame = file_name if isinstance(file_name, pd.DataFrame) else (table if isinstance(table, pd.DataFrame) else None)
        table = table.to_numpy() if isinstance(table, pd.DataFrame) else np.asarray(table)
        if table.ndim != 2:
            table = table.reshape(len(table), -1) if table.size else table.reshape(0, len(headers))
        self._schema = headers
        self._data = table
        self._device = None          # filled by FutharkContext.create_table
        self._frame = frame

    def get_schema(self):
        return self._schema

    def get_data(self):
        return self._data

    def get_name(self):
        return self._table_name

    def _raw_column(self, j):
        c = self._data[:, j] if self._frame is None else self._frame.iloc[:, j].to_numpy()
        if c.dtype == object:
            c = pd.to_numeric(pd.Series(c)).to_numpy()
        return c

    def column_dtypes(self):
        """The device dtype every column would get from THIS table's values (shards of one table agree on the widest,
        see DTYPE_CODES)."""
        return [np.dtype(column_dtype(self._raw_column(j))) for j in range(self._data.shape[1])]

    def host_columns(self, dtypes=None):
        """Per-column contiguous arrays in their device dtype (or in `dtypes`, one per column).  Object-typed
        DataFrame blocks (mixed columns) are converted column by column."""
        cols = []
        for j in range(self._data.shape[1]):
            c = self._raw_column(j)
            cols.append(np.ascontiguousarray(c.astype(column_dtype(c) if dtypes is None else dtypes[j])))
        return cols


# Shards of one table must agree on every column's device dtype although each sees only its own values: int32 < uint32 <
# int64 < float32 as codes, the widest wins (one MAX all-reduce of the codes, dist.ShardedFutharkContext.create_table).
DTYPE_CODES = [np.dtype(np.int32), np.dtype(np.uint32), np.dtype(np.int64), np.dtype(np.float32)]


def dtype_code(dt):
    return DTYPE_CODES.index(np.dtype(dt))


def read_csv_byte_range(file_name, part, parts):
    """The rows of a CSV file whose first byte lies in the part-th of `parts` equal byte ranges of the data section (a
    row belongs to the range its first byte falls in), parsed with the file's header: (DataFrame, headers).  Every row
    of the file belongs to exactly one part and parts follow each other in file order, so part r of R is shard r of a
    row-range-sharded table -- and no process parses more than its share of the file."""
    import io
    import os
    size = os.path.getsize(file_name)
    with open(file_name, "rb") as f:
        header = f.readline()
        data0 = f.tell()
        span = size - data0

        def boundary(i):                       # first byte of the first row that starts at or after data0 + i * span / parts
            if i <= 0:
                return data0
            if i >= parts:
                return size
            at = data0 + (span * i) // parts
            f.seek(at - 1)                      # a row starts at `at` iff the byte before it is a newline
            f.readline()
            return min(f.tell(), size)

        lo, hi = boundary(part), boundary(part + 1)
        f.seek(lo)
        body = f.read(max(hi - lo, 0))
    table = pd.read_csv(io.BytesIO(header + body), skipinitialspace=True)
    headers = [str(h).strip() for h in table.columns.tolist()]
    table.columns = headers
    return table, headers
