"""The boundary FORMAT, pinned with the reference's own C inspector.

oracle/_ref/libcinspect_arrow.so is the reference's tests/c_inspect_arrow.c compiled where it lies (oracle/Makefile).
The arrays its Rust tests export (tests/arrow_c_integration.rs:15-223; vectors in tests/golden/arrow_c_kat.json) are
produced here by PyArrow through the same C Data Interface and must satisfy the same C checks: the struct layout,
buffers[0] = LSB validity, buffers[1] = values, and the format strings this repo's ABI consumes are the
reference's. CPU only."""
import ctypes as C
import json
from pathlib import Path

import pyarrow as pa
import pytest

from minarrow_amd.arrow_c import ArrowArray, ArrowSchema, Exported

KAT = json.loads((Path(__file__).resolve().parent / "golden" / "arrow_c_kat.json").read_text())
REF = Path(__file__).resolve().parent.parent / "oracle" / "_ref" / "libcinspect_arrow.so"
PA_TYPE = {"i": pa.int32(), "l": pa.int64(), "I": pa.uint32(), "f": pa.float32(), "g": pa.float64()}


@pytest.fixture(scope="module")
def inspector():
    if not REF.exists():
        from oracle import oracle

        oracle.build(force=True)
    if not REF.exists():
        pytest.skip("oracle/_ref was not built (reference tree absent and no prebuilt inspector)")
    lib = C.CDLL(str(REF))
    lib.c_arrow_check_schema.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    return lib


def test_struct_layout_matches_the_c_declaration():
    # tests/c_inspect_arrow.c:17-41 — 10 and 9 eight-byte fields
    assert C.sizeof(ArrowArray) == 80 and C.sizeof(ArrowSchema) == 72
    assert ArrowArray.buffers.offset == 40 and ArrowArray.release.offset == 64
    assert ArrowSchema.format.offset == 0 and ArrowSchema.release.offset == 56


@pytest.mark.parametrize("case", KAT["cases"], ids=lambda c: c["checker"])
def test_reference_inspector_accepts_exported_arrays(inspector, case):
    arr = pa.array(case["values"], type=PA_TYPE[case["format"]])
    with Exported(arr) as ex:
        fn = getattr(inspector, case["checker"])
        fn.argtypes = [C.c_void_p]
        assert fn(ex.array_ptr) == 1
        assert ex.schema.format == case["format"].encode()
        ex.schema.name = b"x"
        assert inspector.c_arrow_check_schema(ex.schema_ptr, b"x", case["format"].encode()) == 1
        if "validity_lsb" in case:
            bitmap = C.cast(ex.array.buffers[0], C.POINTER(C.c_uint8))
            assert [(bitmap[i >> 3] >> (i & 7)) & 1 for i in range(len(case["values"]))] == case["validity_lsb"]
            assert ex.array.null_count == 1
