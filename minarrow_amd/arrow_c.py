"""ctypes mirror of the Arrow C Data Interface structs (the layout src/ffi/arrow_c_ffi.rs declares with
#[repr(C)] and tests/c_inspect_arrow.c:17-41 reads from C), plus helpers to obtain them from PyArrow — the same
`_export_to_c` protocol the reference's Python bridge uses (pyo3/src/ffi/to_rust.rs:282)."""
from __future__ import annotations

import ctypes as C


class ArrowSchema(C.Structure):
    pass


class ArrowArray(C.Structure):
    pass


ArrowSchema._fields_ = [
    ("format", C.c_char_p),
    ("name", C.c_char_p),
    ("metadata", C.c_char_p),
    ("flags", C.c_int64),
    ("n_children", C.c_int64),
    ("children", C.POINTER(C.POINTER(ArrowSchema))),
    ("dictionary", C.POINTER(ArrowSchema)),
    ("release", C.c_void_p),
    ("private_data", C.c_void_p),
]

ArrowArray._fields_ = [
    ("length", C.c_int64),
    ("null_count", C.c_int64),
    ("offset", C.c_int64),
    ("n_buffers", C.c_int64),
    ("n_children", C.c_int64),
    ("buffers", C.POINTER(C.c_void_p)),
    ("children", C.POINTER(C.POINTER(ArrowArray))),
    ("dictionary", C.POINTER(ArrowArray)),
    ("release", C.c_void_p),
    ("private_data", C.c_void_p),
]


class Exported:
    """An Arrow array exported through the C Data Interface. Keeps the producer alive and calls the
    producer's release callbacks on close (the consumer side of the protocol)."""

    def __init__(self, pa_array):
        self.array = ArrowArray()
        self.schema = ArrowSchema()
        self._keep = pa_array
        pa_array._export_to_c(C.addressof(self.array), C.addressof(self.schema))

    @property
    def array_ptr(self) -> int:
        return C.addressof(self.array)

    @property
    def schema_ptr(self) -> int:
        return C.addressof(self.schema)

    def close(self) -> None:
        for st in (self.array, self.schema):
            if st.release:
                C.CFUNCTYPE(None, C.c_void_p)(st.release)(C.addressof(st))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Owned:
    """An ArrowArray / ArrowSchema pair PRODUCED by the library (ma_apply_arrow_export, ma_apply_arrow_batch_export).
    Either hand it to a consumer with `to_pyarrow()` (which moves ownership: PyArrow calls `release` when the
    array dies) or call `close()` to run the release callbacks here."""

    def __init__(self):
        self.array = ArrowArray()
        self.schema = ArrowSchema()

    @property
    def array_ptr(self) -> int:
        return C.addressof(self.array)

    @property
    def schema_ptr(self) -> int:
        return C.addressof(self.schema)

    @property
    def released(self) -> bool:
        return not self.array.release and not self.schema.release

    def to_pyarrow(self, record_batch: bool = False):
        import pyarrow as pa

        cls = pa.RecordBatch if record_batch else pa.Array
        return cls._import_from_c(self.array_ptr, self.schema_ptr)  # moves both structs (marks them released)

    def close(self) -> None:
        for st in (self.array, self.schema):
            if st.release:
                C.CFUNCTYPE(None, C.c_void_p)(st.release)(C.addressof(st))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class ArrowArrayStream(C.Structure):
    _fields_ = [
        ("get_schema", C.c_void_p),
        ("get_next", C.c_void_p),
        ("get_last_error", C.c_void_p),
        ("release", C.c_void_p),
        ("private_data", C.c_void_p),
    ]


class ExportedStream:
    """A RecordBatchReader (or anything with `_export_to_c`) exported as an ArrowArrayStream."""

    def __init__(self, reader):
        self.stream = ArrowArrayStream()
        self._keep = reader
        reader._export_to_c(C.addressof(self.stream))

    @property
    def ptr(self) -> int:
        return C.addressof(self.stream)

    def close(self) -> None:
        if self.stream.release:
            C.CFUNCTYPE(None, C.c_void_p)(self.stream.release)(C.addressof(self.stream))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
