"""Packing of a CompiledModel + env configuration into the flat "RSRM" blob.

The blob is the only thing that crosses the C ABI at model-creation time
(`rsr_model_create(const void* blob, size_t nbytes, ...)`, include/rsr_mjx.h):
a header, a table of named fields and the field data, float32 / int32 only.

  header   : char magic[4]="RSRM"; int32 version; int32 nfields; int32 total_bytes
  field[i] : char name[40]; int32 dtype (0=f32, 1=i32); int32 count; int32 offset_bytes; int32 reserved
  data     : each field 16-byte aligned

Both the HIP library and the CPU oracle look fields up by name, so neither
depends on the other's headers.
"""
from __future__ import annotations

import struct
from typing import Dict

import numpy as np

from .mjcf import CompiledModel

MAGIC = b"RSRM"
VERSION = 1
_NAME_LEN = 40
_ENTRY = struct.Struct(f"<{_NAME_LEN}siiii")
_HEADER = struct.Struct("<4siii")

# model arrays that go to the device / oracle (everything numeric)
_SKIP_PREFIX = ("__",)


def pack_blob(fields: Dict[str, np.ndarray]) -> bytes:
    names = sorted(fields)
    table_bytes = _HEADER.size + _ENTRY.size * len(names)
    off = (table_bytes + 15) & ~15
    entries, chunks = [], []
    for n in names:
        a = np.ascontiguousarray(fields[n])
        if a.dtype.kind == "f":
            a = a.astype(np.float32)
            dt = 0
        elif a.dtype.kind in "iub":
            a = a.astype(np.int32)
            dt = 1
        else:
            raise TypeError(f"field {n}: unsupported dtype {a.dtype}")
        raw = a.tobytes()
        if len(n.encode()) >= _NAME_LEN:
            raise ValueError(f"field name too long: {n}")
        entries.append(_ENTRY.pack(n.encode(), dt, a.size, off, 0))
        pad = (-len(raw)) & 15
        chunks.append(raw + b"\0" * pad)
        off += len(raw) + pad
    head = _HEADER.pack(MAGIC, VERSION, len(names), off)
    body = head + b"".join(entries)
    body += b"\0" * (((table_bytes + 15) & ~15) - len(body))
    return body + b"".join(chunks)


def unpack_blob(blob: bytes) -> Dict[str, np.ndarray]:
    magic, ver, n, total = _HEADER.unpack_from(blob, 0)
    if magic != MAGIC or ver != VERSION:
        raise ValueError("not an RSRM v1 blob")
    out = {}
    for i in range(n):
        name, dt, cnt, off, _ = _ENTRY.unpack_from(blob, _HEADER.size + i * _ENTRY.size)
        name = name.rstrip(b"\0").decode()
        dtype = np.float32 if dt == 0 else np.int32
        out[name] = np.frombuffer(blob, dtype=dtype, count=cnt, offset=off).copy()
    return out


def model_fields(m: CompiledModel) -> Dict[str, np.ndarray]:
    f = {}
    for k, v in m.arrays.items():
        if k.startswith(_SKIP_PREFIX):
            continue
        f[k] = np.asarray(v)
    f["dims"] = np.array([m.nq, m.nv, m.nu, m.nbody, m.njnt, m.ngeom, m.nsite,
                          int(m.arrays["eq_obj1id"].shape[0]), m.npair], dtype=np.int32)
    return f
