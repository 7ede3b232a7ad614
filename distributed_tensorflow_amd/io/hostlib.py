"""ctypes signatures for ``libdtfa_host.so`` (file formats, JPEG entropy decoder, record loader)."""
from __future__ import annotations

import ctypes as C

from .. import _lib

_done = False
vp, cp, i32, i64, u64, u32 = C.c_void_p, C.c_char_p, C.c_int, C.c_longlong, C.c_uint64, C.c_uint32
pp_u8 = C.POINTER(C.POINTER(C.c_uint8))


class JpegInfo(C.Structure):
    _fields_ = [
        ("width", i32), ("height", i32), ("ncomp", i32),
        ("hs", i32 * 4), ("vs", i32 * 4),
        ("blocks_w", i32 * 4), ("blocks_h", i32 * 4),
        ("coef_offset", i64 * 4), ("total_coefs", i64),
        ("qt", (C.c_uint16 * 64) * 4),
        ("hmax", i32), ("vmax", i32), ("restart_interval", i32), ("sof", i32),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("slot", i32), ("count", i32),
        ("offsets", C.POINTER(u64)), ("heights", C.POINTER(i32)), ("widths", C.POINTER(i32)),
        ("channels", C.POINTER(i32)), ("labels", C.POINTER(i32)), ("bytes_used", u64),
    ]


_SIGS = {
    "dtfa_crc32c": ([vp, C.c_size_t], u32),
    "dtfa_crc32c_extend": ([u32, vp, C.c_size_t], u32),
    "dtfa_crc32c_masked": ([vp, C.c_size_t], u32),
    "dtfa_crc32c_sw_for_test": ([vp, C.c_size_t], u32),
    "dtfa_tfrecord_writer_open": ([cp, i32], vp),
    "dtfa_tfrecord_write": ([vp, vp, u64], i32),
    "dtfa_tfrecord_writer_flush": ([vp], i32),
    "dtfa_tfrecord_writer_close": ([vp], i32),
    "dtfa_tfrecord_reader_open": ([cp, i32], vp),
    "dtfa_tfrecord_reader_seek": ([vp, u64], i32),
    "dtfa_tfrecord_read_next": ([vp, pp_u8, C.POINTER(u64)], i32),
    "dtfa_tfrecord_reader_close": ([vp], i32),
    "dtfa_tfrecord_index": ([cp, C.POINTER(u64), C.POINTER(u64), i64], i64),
    "dtfa_example_builder_new": ([], vp),
    "dtfa_example_builder_free": ([vp], None),
    "dtfa_example_add_bytes": ([vp, cp, vp, u64], None),
    "dtfa_example_add_int64": ([vp, cp, C.POINTER(i64), i32], None),
    "dtfa_example_add_float": ([vp, cp, C.POINTER(C.c_float), i32], None),
    "dtfa_example_serialize": ([vp, pp_u8, C.POINTER(u64)], i32),
    "dtfa_example_parse": ([vp, u64], vp),
    "dtfa_example_free": ([vp], None),
    "dtfa_example_num_features": ([vp], i32),
    "dtfa_example_feature_info": ([vp, cp, C.POINTER(i32)], i32),
    "dtfa_example_get_bytes": ([vp, cp, i32, pp_u8, C.POINTER(u64)], i32),
    "dtfa_example_get_int64": ([vp, cp, C.POINTER(i64), i32], i32),
    "dtfa_example_get_float": ([vp, cp, C.POINTER(C.c_float), i32], i32),
    "dtfa_example_feature_name": ([vp, i32], cp),
    "dtfa_bundle_writer_new": ([cp], vp),
    "dtfa_bundle_add": ([vp, cp, i32, C.POINTER(i64), i32, vp, u64], i32),
    "dtfa_bundle_writer_finish": ([vp], i32),
    "dtfa_bundle_writer_abort": ([vp], None),
    "dtfa_bundle_reader_open": ([cp], vp),
    "dtfa_bundle_reader_error": ([vp], cp),
    "dtfa_bundle_num_entries": ([vp], i32),
    "dtfa_bundle_num_shards": ([vp], i32),
    "dtfa_bundle_entry": ([vp, i32, C.POINTER(cp), C.POINTER(i32), C.POINTER(i64), C.POINTER(i32), C.POINTER(u64),
                           C.POINTER(u64), C.POINTER(u32)], i32),
    "dtfa_bundle_reader_close": ([vp], None),
    "dtfa_table_write": ([cp, i32, C.POINTER(cp), C.POINTER(u64), C.POINTER(cp), C.POINTER(u64), i32], i32),
    "dtfa_table_read": ([cp], vp),
    "dtfa_table_error": ([vp], cp),
    "dtfa_table_count": ([vp], i32),
    "dtfa_table_get": ([vp, i32, C.POINTER(vp), C.POINTER(u64), C.POINTER(vp), C.POINTER(u64)], i32),
    "dtfa_table_free": ([vp], None),
    "dtfa_events_open": ([cp, cp], vp),
    "dtfa_events_path": ([vp], cp),
    "dtfa_events_begin": ([vp, i64, C.c_double], None),
    "dtfa_events_add_scalar": ([vp, cp, C.c_float], None),
    "dtfa_events_add_histogram": ([vp, cp, C.POINTER(C.c_float), i64], None),
    "dtfa_events_end": ([vp], i32),
    "dtfa_events_flush": ([vp], i32),
    "dtfa_events_close": ([vp], i32),
    "dtfa_jpeg_parse_header": ([vp, u64, C.POINTER(JpegInfo)], i32),
    "dtfa_jpeg_decode_coefs": ([vp, u64, C.POINTER(JpegInfo), vp, i64], i32),
    "dtfa_jpeg_info_size": ([], i32),
    "dtfa_loader_open": ([cp, i32, i32, i32, i32, i32, C.POINTER(vp), u64, u64, i32, i32, i32], vp),
    "dtfa_loader_num_records": ([vp], i64),
    "dtfa_loader_num_batches": ([vp], i64),
    "dtfa_loader_next": ([vp, C.POINTER(Batch)], i32),
    "dtfa_loader_release": ([vp, i32], None),
    "dtfa_loader_close": ([vp], None),
}


def lib():
    global _done
    L = _lib.host()
    if not _done:
        for name, (args, res) in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res
        if L.dtfa_jpeg_info_size() != C.sizeof(JpegInfo):
            raise RuntimeError("JpegInfo layout mismatch between Python and libdtfa_host.so")
        _done = True
    return L
