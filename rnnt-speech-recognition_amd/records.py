"""The reference's on-disk training-set format, read and written without TensorFlow.

The reference stores each example as a `tf.train.Example` with five bytes features -- `mel_specs`, `pred_inp`,
`spec_lengths`, `label_lengths`, `labels` -- every one a `tf.io.serialize_tensor` TensorProto
(utils/preprocessing.py:110-161), in `<name>.tfrecord` files written by `tf.data.experimental.TFRecordWriter`
(preprocess_common_voice.py:24-30) and listed by `load_dataset` (utils/preprocessing.py:97-107).  This module speaks
that format directly: TFRecord framing (length, masked CRC-32C of the length, payload, masked CRC-32C of the payload),
the protobuf wire format of Example / Features / Feature / BytesList, and the TensorProto fields serialize_tensor emits
(dtype, tensor_shape, tensor_content; the repeated *_val fields are accepted too).  The tuples it yields are the ones
`features.make_record` builds and `features.padded_batch` batches.
"""
from __future__ import annotations

import glob
import os
import struct
from typing import Dict, Iterable, Iterator, List, Sequence, Tuple

import numpy as np
import torch

FEATURE_KEYS = ("mel_specs", "pred_inp", "spec_lengths", "label_lengths", "labels")

# ---------------------------------------------------------------------------------------------- CRC-32C + framing
_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (0x82F63B78 if _c & 1 else 0)
    _CRC_TABLE.append(_c)
_MASK_DELTA = 0xA282EAD8


def crc32c(data: bytes) -> int:
    """CRC-32C (Castagnoli), the checksum of the TFRecord framing."""
    c = 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def read_tfrecord_payloads(path: str, verify_payload: bool = False) -> Iterator[bytes]:
    """The payloads of one TFRecord file.  The 12-byte header (length + its checksum) is always verified; the payload
    checksum only on request (a pure-Python CRC over a spectrogram costs tens of milliseconds)."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise ValueError(f"{path}: truncated record header")
            (length,) = struct.unpack("<Q", head[:8])
            if struct.unpack("<I", head[8:])[0] != masked_crc32c(head[:8]):
                raise ValueError(f"{path}: corrupt record length")
            body = f.read(length + 4)
            if len(body) != length + 4:
                raise ValueError(f"{path}: truncated record")
            payload = body[:length]
            if verify_payload and struct.unpack("<I", body[length:])[0] != masked_crc32c(payload):
                raise ValueError(f"{path}: corrupt record payload")
            yield payload


def frame_tfrecord(payload: bytes) -> bytes:
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", masked_crc32c(head)) + payload + struct.pack("<I", masked_crc32c(payload))


# ---------------------------------------------------------------------------------------------- protobuf wire format
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint longer than 64 bits")


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) of one message; value is an int (varint, fixed) or bytes (length-delimited)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = struct.unpack_from("<Q", buf, pos)[0], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise ValueError("truncated length-delimited field")
            val, pos = buf[pos : pos + ln], pos + ln
        elif wt == 5:
            val, pos = struct.unpack_from("<I", buf, pos)[0], pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield num, wt, val


def _put_varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _put_bytes(num: int, payload: bytes) -> bytes:
    return _put_varint((num << 3) | 2) + _put_varint(len(payload)) + payload


# ---------------------------------------------------------------------------------------------- TensorProto
# tensorflow/core/framework/types.proto; (numpy dtype, field number of the repeated *_val fallback)
_DTYPES = {1: (np.float32, 5), 2: (np.float64, 6), 3: (np.int32, 7), 4: (np.uint8, 7), 5: (np.int16, 7),
           6: (np.int8, 7), 9: (np.int64, 10), 10: (np.bool_, 11), 19: (np.float16, 13)}
_DTYPE_CODE = {np.dtype(v[0]): k for k, v in _DTYPES.items()}


def parse_tensor(buf: bytes) -> np.ndarray:
    """tf.io.parse_tensor for numeric tensors."""
    dtype_code, dims, content, vals = 0, [], None, {}
    for num, wt, val in _fields(buf):
        if num == 1:
            dtype_code = val
        elif num == 2:
            for n2, _, dim in _fields(val):
                if n2 == 2:
                    size = 0
                    for n3, _, v in _fields(dim):
                        if n3 == 1:
                            size = v - (1 << 64) if v >> 63 else v
                    dims.append(size)
                elif n2 == 3 and dim:
                    raise ValueError("tensor of unknown rank")
        elif num == 4:
            content = val
        elif num in (5, 6, 7, 10, 11, 13):
            vals.setdefault(num, []).append((wt, val))
    if dtype_code not in _DTYPES:
        raise ValueError(f"unsupported tensor dtype enum {dtype_code}")
    np_dtype, val_field = _DTYPES[dtype_code]
    if any(d < 0 for d in dims):
        raise ValueError("tensor with unknown dimension")
    count = int(np.prod(dims, dtype=np.int64)) if dims else 1
    if content is not None:
        arr = np.frombuffer(content, dtype=np.dtype(np_dtype).newbyteorder("<"))
        if arr.size != count:
            raise ValueError(f"tensor_content holds {arr.size} elements, shape {dims} needs {count}")
        return arr.astype(np_dtype).reshape(dims)
    flat: List = []
    for wt, val in vals.get(val_field, []):
        if wt == 2 and val_field == 5:
            flat.extend(np.frombuffer(val, "<f4").tolist())
        elif wt == 2 and val_field == 6:
            flat.extend(np.frombuffer(val, "<f8").tolist())
        elif wt == 2:  # packed varints
            pos = 0
            while pos < len(val):
                v, pos = _varint(val, pos)
                flat.append(v - (1 << 64) if v >> 63 else v)
        elif wt == 5:
            flat.append(struct.unpack("<f", struct.pack("<I", val))[0])
        elif wt == 1:
            flat.append(struct.unpack("<d", struct.pack("<Q", val))[0])
        else:
            flat.append(val - (1 << 64) if val >> 63 else val)
    if np_dtype is np.float16:
        flat = np.array(flat, dtype=np.uint16).view(np.float16).tolist()
    if not flat:
        flat = [0]
    if len(flat) < count:  # the *_val fields may be truncated: the last value repeats
        flat = flat + [flat[-1]] * (count - len(flat))
    return np.array(flat[:count], dtype=np_dtype).reshape(dims)


def serialize_tensor(arr) -> bytes:
    """tf.io.serialize_tensor for numeric tensors: dtype, shape, little-endian tensor_content."""
    arr = np.asarray(arr)
    if arr.dtype not in _DTYPE_CODE:
        raise ValueError(f"unsupported dtype {arr.dtype}")
    shape = b"".join(_put_bytes(2, _put_varint(1 << 3) + _put_varint(d)) for d in arr.shape)
    content = np.ascontiguousarray(arr).astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()
    return _put_varint(1 << 3) + _put_varint(_DTYPE_CODE[arr.dtype]) + _put_bytes(2, shape) + _put_bytes(4, content)


# ---------------------------------------------------------------------------------------------- tf.train.Example
def parse_bytes_features(buf: bytes) -> Dict[str, List[bytes]]:
    """Example{features=1: Features{feature=1: map<string, Feature{bytes_list=1: BytesList{value=1}}>}}.  Features of
    the other two kinds (float_list, int64_list) are skipped: the reference writes none."""
    out: Dict[str, List[bytes]] = {}
    for num, wt, features in _fields(buf):
        if num != 1 or wt != 2:
            continue
        for n2, w2, entry in _fields(features):
            if n2 != 1 or w2 != 2:
                continue
            key, feature = None, b""
            for n3, _, v in _fields(entry):
                if n3 == 1:
                    key = v.decode("utf8")
                elif n3 == 2:
                    feature = v
            for n4, w4, kind in _fields(feature):
                if n4 == 1 and w4 == 2:
                    out[key] = [v for n5, _, v in _fields(kind) if n5 == 1]
    return out


def serialize_example(mel_specs, pred_inp, spec_lengths, label_lengths, labels) -> bytes:
    """utils/preprocessing.py:133-161: five single-element bytes lists, each a serialized tensor."""
    host = lambda x: x.detach().cpu() if isinstance(x, torch.Tensor) else x  # records computed on the device are fine
    tensors = (np.asarray(host(mel_specs), np.float32), np.asarray(host(pred_inp), np.int32),
               np.asarray(host(spec_lengths), np.int32), np.asarray(host(label_lengths), np.int32),
               np.asarray(host(labels), np.int32))
    entries = b""
    for key, t in zip(FEATURE_KEYS, tensors):
        feature = _put_bytes(1, _put_bytes(1, serialize_tensor(t)))
        entries += _put_bytes(1, _put_bytes(1, key.encode()) + _put_bytes(2, feature))
    return _put_bytes(1, entries)


def parse_example(payload: bytes):
    """utils/preprocessing.py:110-130, to the tuple layout of features.make_record."""
    feats = parse_bytes_features(payload)
    missing = [k for k in FEATURE_KEYS if not feats.get(k)]
    if missing:
        raise ValueError(f"example lacks feature(s) {missing}")
    mel, pred_inp, spec_len, label_len, labels = (parse_tensor(feats[k][0]) for k in FEATURE_KEYS)
    if mel.ndim != 2 or pred_inp.ndim != 1 or labels.ndim != 1 or spec_len.ndim or label_len.ndim:
        raise ValueError("example tensors do not have the ranks ([T,F], [U], [], [], [L]) of the reference's records")
    return (torch.from_numpy(mel.astype(np.float32)), torch.from_numpy(pred_inp.astype(np.int32)), int(spec_len),
            int(label_len), torch.from_numpy(labels.astype(np.int32)))


# ---------------------------------------------------------------------------------------------- datasets
def load_dataset(data_dir: str, name: str, verify_payload: bool = False):
    """utils/preprocessing.py:97-107: the examples of `<data_dir>/<name>.tfrecord` (name may hold a glob pattern)."""
    for path in sorted(glob.glob(os.path.join(data_dir, f"{name}.tfrecord"))):
        for payload in read_tfrecord_payloads(path, verify_payload):
            yield parse_example(payload)


def write_dataset(records: Iterable[Sequence], path: str) -> int:
    """preprocess_common_voice.py:24-30: one TFRecord file of serialized examples; returns the number written."""
    n = 0
    with open(path, "wb") as f:
        for rec in records:
            f.write(frame_tfrecord(serialize_example(*rec)))
            n += 1
    return n


def batches(records: Iterable, batch_size: int, max_size: int = None, drop_remainder: bool = False):
    """run_rnnt.py:66-91 get_dataset: take(max_size) -> padded_batch(batch_size)."""
    from .features import padded_batch

    pending = []
    for i, rec in enumerate(records):
        if max_size is not None and i >= max_size:
            break
        pending.append(rec)
        if len(pending) == batch_size:
            yield padded_batch(pending)
            pending = []
    if pending and not drop_remainder:
        yield padded_batch(pending)
