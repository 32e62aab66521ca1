"""Input side of the path: the S3DIS block records the reference trains from, and the per-step block
sampling / augmentation of its training loop — without TensorFlow.

  * record files: TFRecord framing (length, masked CRC-32C of the length, payload, masked CRC-32C of the payload) around a
    serialized ``tf.train.Example`` whose features are raw little-endian arrays — ``xyz_raw`` / ``rel_xyz_raw`` /
    ``rgb_raw`` float32 [n,3], ``seg_label`` / ``inner_label`` / ``index_label`` int32 [n], ``scene_label`` /
    ``scene_idx`` int64 scalars (io/make_tfrecord_s3dis.py:227-242).  The protobuf wire format of Example is three nested
    length-delimited messages and a map; it is encoded / decoded here by hand (tests check it against google.protobuf
    with the same schema);
  * ``parse_block``: what parse_fn builds (s3dis_seg/train_s3dis.py:145-171): [n, 8] = xyz, rgb, label, inner;
  * ``sample_points``: NUM_POINT rows per block, without replacement when the block has enough points (:343-347);
  * ``augment_batch``: shuffle the blocks and the point order, rotate about z + small random rotation on the first third
    of the batch, jitter the second third (:124-141, utils/data_util.py:47-61,140-176).

There is no S3DIS data in this environment; bench.py keeps its synthetic generator (harness/synth.py) and these
functions are exercised by round-trip / known-answer tests (tests/test_blockio.py).
"""
import struct

import numpy as np

# ---------------------------------------------------------------------------------------------------------------
# CRC-32C (Castagnoli), table driven; TFRecord stores it "masked"
# ---------------------------------------------------------------------------------------------------------------
_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE.append(_c)
_TABLE_NP = np.array(_TABLE, dtype=np.uint32)


def crc32c(data):
    crc = 0xFFFFFFFF
    tab = _TABLE
    for b in bytes(data):
        crc = tab[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------------------
# protobuf wire format, the subset tf.train.Example needs
# ---------------------------------------------------------------------------------------------------------------
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _ld(field, payload):          # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _feature(value):
    """Feature{ bytes_list = 1 | float_list = 2 | int64_list = 3 }"""
    if isinstance(value, (bytes, bytearray)):
        return _ld(1, _ld(1, bytes(value)))
    arr = np.asarray(value)
    if arr.dtype.kind == "f":
        return _ld(2, _ld(1, arr.astype("<f4").tobytes()))                       # packed floats
    return _ld(3, _ld(1, b"".join(_varint(int(v)) for v in arr.reshape(-1))))   # packed varints


def encode_example(features):
    """dict name -> bytes | float array | int array  ->  serialized tf.train.Example"""
    entries = b""
    for name in sorted(features):        # deterministic output (protobuf map order is unspecified)
        entry = _ld(1, name.encode("utf-8")) + _ld(2, _feature(features[name]))
        entries += _ld(1, entry)         # Features.feature map entry
    return _ld(1, entries)               # Example.features


def _fields(buf):
    pos = 0
    while pos < len(buf):
        tag, pos = _read_varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 2:
            n, pos = _read_varint(buf, pos)
            yield field, wire, buf[pos:pos + n]
            pos += n
        elif wire == 0:
            v, pos = _read_varint(buf, pos)
            yield field, wire, v
        elif wire == 5:
            yield field, wire, buf[pos:pos + 4]
            pos += 4
        elif wire == 1:
            yield field, wire, buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wire)


def _decode_feature(buf):
    for field, _w, payload in _fields(buf):
        if field == 1:                                        # BytesList
            vals = [bytes(p) for f, _w2, p in _fields(payload) if f == 1]
            return vals[0] if len(vals) == 1 else vals
        if field == 2:                                        # FloatList (packed or not)
            out = []
            for f, w, p in _fields(payload):
                if f == 1:
                    out.append(np.frombuffer(bytes(p), dtype="<f4"))
            return np.concatenate(out) if out else np.zeros(0, np.float32)
        if field == 3:                                        # Int64List
            out = []
            for f, w, p in _fields(payload):
                if f != 1:
                    continue
                if w == 0:
                    out.append(p)
                else:
                    pos = 0
                    while pos < len(p):
                        v, pos = _read_varint(p, pos)
                        out.append(v)
            return np.array([v - (1 << 64) if v >= (1 << 63) else v for v in out], dtype=np.int64)
    return None


def decode_example(buf):
    """serialized tf.train.Example -> dict name -> bytes | float32 array | int64 array"""
    out = {}
    buf = memoryview(bytes(buf))
    for field, _w, feats in _fields(buf):
        if field != 1:
            continue
        for f2, _w2, entry in _fields(feats):
            if f2 != 1:
                continue
            key, val = None, None
            for f3, _w3, p in _fields(entry):
                if f3 == 1:
                    key = bytes(p).decode("utf-8")
                elif f3 == 2:
                    val = _decode_feature(p)
            out[key] = val
    return out


# ---------------------------------------------------------------------------------------------------------------
# TFRecord files
# ---------------------------------------------------------------------------------------------------------------
def write_records(path, payloads):
    with open(path, "wb") as f:
        for data in payloads:
            head = struct.pack("<Q", len(data))
            f.write(head)
            f.write(struct.pack("<I", masked_crc32c(head)))
            f.write(data)
            f.write(struct.pack("<I", masked_crc32c(data)))


def read_records(path, verify=True):
    with open(path, "rb") as f:
        while True:
            head = f.read(8)
            if not head:
                return
            if len(head) != 8:
                raise IOError("truncated record header in %s" % path)
            (n,) = struct.unpack("<Q", head)
            (hc,) = struct.unpack("<I", f.read(4))
            if verify and hc != masked_crc32c(head):
                raise IOError("corrupt record length in %s" % path)
            data = f.read(n)
            (dc,) = struct.unpack("<I", f.read(4))
            if len(data) != n or (verify and dc != masked_crc32c(data)):
                raise IOError("corrupt record in %s" % path)
            yield data


# ---------------------------------------------------------------------------------------------------------------
# S3DIS blocks
# ---------------------------------------------------------------------------------------------------------------
def encode_block(xyz, rgb, seg_label, inner_label, rel_xyz=None, index_label=None, scene_label=0, scene_idx=0):
    """one block record with the feature names and raw layouts of io/make_tfrecord_s3dis.py:227-242"""
    xyz = np.ascontiguousarray(xyz, dtype="<f4")
    n = xyz.shape[0]
    rel_xyz = xyz if rel_xyz is None else rel_xyz
    index_label = np.arange(n) if index_label is None else index_label
    return encode_example({
        "xyz_raw": xyz.tobytes(),
        "rel_xyz_raw": np.ascontiguousarray(rel_xyz, dtype="<f4").tobytes(),
        "rgb_raw": np.ascontiguousarray(rgb, dtype="<f4").tobytes(),
        "seg_label": np.ascontiguousarray(seg_label, dtype="<i4").tobytes(),
        "inner_label": np.ascontiguousarray(inner_label, dtype="<i4").tobytes(),
        "index_label": np.ascontiguousarray(index_label, dtype="<i4").tobytes(),
        "scene_label": np.array([scene_label], dtype=np.int64),
        "scene_idx": np.array([scene_idx], dtype=np.int64),
    })


def parse_block(record):
    """-> float32 [n, 8]: xyz, rgb, label, inner — the tensor parse_fn hands to the training loop (train_s3dis.py:145-171)"""
    ex = decode_example(record)
    xyz = np.frombuffer(ex["xyz_raw"], dtype="<f4").reshape(-1, 3)
    rgb = np.frombuffer(ex["rgb_raw"], dtype="<f4").reshape(-1, 3)
    seg = np.frombuffer(ex["seg_label"], dtype="<i4").reshape(-1, 1).astype(np.float32)
    inner = np.frombuffer(ex["inner_label"], dtype="<i4").reshape(-1, 1).astype(np.float32)
    if not (len(xyz) == len(rgb) == len(seg) == len(inner)):
        raise ValueError("block record with inconsistent array lengths")
    return np.concatenate((xyz, rgb, seg, inner), axis=1)


def sample_points(block, num_point, rng):
    """NUM_POINT rows of one block: without replacement when it has enough points, else with (train_s3dis.py:343-347)
    -> input [num_point, 6], label [num_point] i32, inner [num_point] i32"""
    n = block.shape[0]
    if n == 0:
        raise ValueError("empty block")
    idx = rng.choice(n, num_point, replace=n < num_point)
    return block[idx, 0:6].astype(np.float32), block[idx, 6].astype(np.int32), block[idx, 7].astype(np.int32)


def _rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]).astype(np.float32)       # float32, like data_util.rot_z (:226-232)


def rotate_point_cloud(batch_xyz, rng, max_angle=2 * np.pi):
    """utils/data_util.py:47-61: every cloud turned about the up (z) axis by its own uniform angle; float32 result.
    One rng.uniform() per cloud, in order — the same draws as the reference makes from np.random."""
    out = np.zeros(batch_xyz.shape, dtype=np.float32)
    for k in range(batch_xyz.shape[0]):
        out[k, ...] = np.dot(batch_xyz[k, ...].reshape((-1, 3)), _rot_z(rng.uniform() * max_angle))
    return out


def rotate_perturbation_point_cloud(batch_xyz, rng, angle_sigma=0.06, angle_clip=0.18):
    """utils/data_util.py:140-163: a small random rotation Rz Ry Rx per cloud (three clipped normal angles); float32 result"""
    out = np.zeros(batch_xyz.shape, dtype=np.float32)
    for k in range(batch_xyz.shape[0]):
        ax, ay, az = np.clip(angle_sigma * rng.randn(3), -angle_clip, angle_clip)
        rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
        ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
        rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
        out[k, ...] = np.dot(batch_xyz[k, ...].reshape((-1, 3)), np.dot(rz, np.dot(ry, rx)))
    return out


def jitter_point_cloud(batch_xyz, rng, sigma=0.01, clip=0.02):
    """utils/data_util.py:166-176: per-point normal noise clipped at `clip`; float64 result (noise + data)"""
    noise = np.clip(sigma * rng.randn(*batch_xyz.shape), -1 * clip, clip)
    noise += batch_xyz
    return noise


def augment_batch(batch_input, batch_label, batch_inner, rng):
    """train_s3dis.py:114-142, draw for draw: shuffle the blocks of the batch, shuffle the point order (the same permutation
    for every block), turn the first third about z by a uniform angle and then by a small random rotation, jitter the
    second third (sigma 0.01, clipped at 0.02); colours, labels and the last third are left alone.  With
    rng = np.random.RandomState(s) the result equals the reference's under np.random.seed(s), bit for bit
    (tests/golden/blockio_ref.npz); the arithmetic runs in the reference's precisions (float64 batch, float32 rotations)
    and the float32 batch the training step consumes is the final cast."""
    bsize, num_point, _ = batch_input.shape
    order = rng.permutation(bsize)                      # == np.random.shuffle(np.arange(bsize))
    batch_input, batch_label, batch_inner = batch_input[order], batch_label[order], batch_inner[order]
    perm = rng.permutation(num_point)
    batch_input, batch_label, batch_inner = batch_input[:, perm], batch_label[:, perm], batch_inner[:, perm]
    batch_input = np.array(batch_input, dtype=np.float64, copy=True)      # the training loop's batch is float64 (:328)
    third = int(np.int32(1 / 3.0 * bsize))
    xyz = rotate_point_cloud(batch_input[0:third, :, 0:3], rng)
    xyz = rotate_perturbation_point_cloud(xyz, rng)
    batch_input[0:third, :, 0:3] = xyz
    batch_input[third:2 * third, :, 0:3] = jitter_point_cloud(batch_input[third:2 * third, :, 0:3], rng)
    return batch_input.astype(np.float32), batch_label, batch_inner


def training_batches(paths, batch_size, num_point, rng, augment=True, shuffle_buffer=10000):
    """one epoch: records of all files, shuffled through a buffer (tf.data's shuffle(buffer_size), :176), NUM_POINT points
    per block, batches of `batch_size` blocks (the last one may be smaller), augmented -> (input [b,n,6] f32, label, inner)"""
    def records():
        for p in paths:
            for r in read_records(p):
                yield r

    buf, it = [], records()
    inputs, labels, inners = [], [], []

    def flush():
        x, l, i = np.stack(inputs), np.stack(labels), np.stack(inners)
        inputs.clear(); labels.clear(); inners.clear()
        return augment_batch(x, l, i, rng) if augment else (x, l, i)

    done = False
    while not done or buf:
        while not done and len(buf) < shuffle_buffer:
            try:
                buf.append(next(it))
            except StopIteration:
                done = True
        if not buf:
            break
        j = rng.randint(len(buf))
        buf[j], buf[-1] = buf[-1], buf[j]
        x, l, i = sample_points(parse_block(buf.pop()), num_point, rng)
        inputs.append(x); labels.append(l); inners.append(i)
        if len(inputs) == batch_size:
            yield flush()
    if inputs:
        yield flush()
