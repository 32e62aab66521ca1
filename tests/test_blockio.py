"""Input side of the path: TFRecord framing + tf.train.Example wire format of the S3DIS block records
(io/make_tfrecord_s3dis.py:227-242), parse_fn's tensor (train_s3dis.py:145-171), block sampling and augmentation
(:116-141, :331-358) — checked against CRC known answers, an independent CRC, google.protobuf with the same schema, round
trips and the invariants of the augmentation."""
import os

import numpy as np
import pytest

from sph3d_gcn_amd.harness import blockio, synth


def _bitwise_crc32c(data):
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def test_crc32c_known_answers():
    assert blockio.crc32c(b"123456789") == 0xE3069283            # the check value of CRC-32C (Castagnoli)
    assert blockio.crc32c(b"") == 0
    assert blockio.crc32c(bytes(32)) == 0x8A9136AA               # RFC 3720 B.4: 32 bytes of zeros
    assert blockio.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43      # RFC 3720 B.4: 32 bytes of 0xFF
    rng = np.random.RandomState(0)
    for n in (1, 7, 64, 1000):
        d = rng.bytes(n)
        assert blockio.crc32c(d) == _bitwise_crc32c(d)
    c = blockio.crc32c(b"abc")
    assert blockio.masked_crc32c(b"abc") == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _example_schema():
    """tf.train.Example's schema (tensorflow/core/example/{example,feature}.proto field numbers) built at run time"""
    pb = pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="sph3d_test_example.proto", package="sph3dtest", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, typ, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=label)
            if tname:
                f.type_name = ".sph3dtest." + tname
        return m

    msg("BytesList", [("value", 1, T.TYPE_BYTES, T.LABEL_REPEATED, None)])
    msg("FloatList", [("value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED, None)])
    msg("Int64List", [("value", 1, T.TYPE_INT64, T.LABEL_REPEATED, None)])
    f = msg("Feature", [("bytes_list", 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, "BytesList"),
                        ("float_list", 2, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, "FloatList"),
                        ("int64_list", 3, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, "Int64List")])
    f.oneof_decl.add(name="kind")
    for fld in f.field:
        fld.oneof_index = 0
    feats = msg("Features", [("feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, "Features.FeatureEntry")])
    entry = feats.nested_type.add(name="FeatureEntry")
    entry.options.map_entry = True
    entry.field.add(name="key", number=1, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    entry.field.add(name="value", number=2, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name=".sph3dtest.Feature")
    msg("Example", [("features", 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, "Features")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("sph3dtest.Example"))


def test_example_wire_format_against_google_protobuf():
    Example = _example_schema()
    rng = np.random.RandomState(1)
    xyz = rng.rand(50, 3).astype(np.float32)
    lab = rng.randint(0, 13, 50).astype(np.int32)
    ours = blockio.encode_example({"xyz_raw": xyz.tobytes(), "seg_label": lab.tobytes(),
                                   "scene_idx": np.array([123456789012, -5], np.int64),
                                   "scale": np.array([0.5, -2.25], np.float32)})
    ex = Example()
    ex.ParseFromString(ours)                                       # google.protobuf reads what we write
    f = ex.features.feature
    assert f["xyz_raw"].bytes_list.value[0] == xyz.tobytes() and f["seg_label"].bytes_list.value[0] == lab.tobytes()
    assert list(f["scene_idx"].int64_list.value) == [123456789012, -5]
    assert list(f["scale"].float_list.value) == [0.5, -2.25]
    back = blockio.decode_example(ex.SerializeToString())          # and we read what google.protobuf writes
    assert back["xyz_raw"] == xyz.tobytes() and back["seg_label"] == lab.tobytes()
    np.testing.assert_array_equal(back["scene_idx"], [123456789012, -5])
    np.testing.assert_array_equal(back["scale"], np.array([0.5, -2.25], np.float32))


def _blocks(n_blocks, rng):
    out = []
    for b in range(n_blocks):
        n = int(rng.randint(3000, 12000))
        xyz, label, inner = synth.s3dis_block(b, n)
        rgb = rng.rand(n, 3).astype(np.float32)
        out.append((xyz, rgb, label.astype(np.int32), inner.astype(np.int32)))
    return out


def test_block_records_round_trip_and_corruption(tmp_path):
    rng = np.random.RandomState(2)
    blocks = _blocks(3, rng)
    path = os.path.join(tmp_path, "Area_1_test.tfrecord")
    blockio.write_records(path, [blockio.encode_block(x, c, l, i, scene_label=4, scene_idx=k)
                                 for k, (x, c, l, i) in enumerate(blocks)])
    recs = list(blockio.read_records(path))
    assert len(recs) == 3
    for rec, (x, c, l, i) in zip(recs, blocks):
        t = blockio.parse_block(rec)
        assert t.dtype == np.float32 and t.shape == (len(x), 8)
        np.testing.assert_array_equal(t[:, 0:3], x)
        np.testing.assert_array_equal(t[:, 3:6], c)
        np.testing.assert_array_equal(t[:, 6], l.astype(np.float32))
        np.testing.assert_array_equal(t[:, 7], i.astype(np.float32))
        ex = blockio.decode_example(rec)
        assert set(ex) == {"xyz_raw", "rel_xyz_raw", "rgb_raw", "seg_label", "inner_label", "index_label", "scene_label",
                           "scene_idx"}
        assert int(ex["scene_label"][0]) == 4
    raw = bytearray(open(path, "rb").read())
    raw[40] ^= 0x01                                                # flip a payload bit: the CRC must catch it
    bad = os.path.join(tmp_path, "bad.tfrecord")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(IOError):
        list(blockio.read_records(bad))
    assert len(list(blockio.read_records(bad, verify=False))) == 3


def test_sampling_and_augmentation_invariants():
    rng = np.random.RandomState(3)
    xyz, label, inner = synth.s3dis_block(0, 9000)
    block = np.concatenate((xyz, rng.rand(9000, 3).astype(np.float32), label[:, None].astype(np.float32), inner[:, None]), 1)
    x, l, i = blockio.sample_points(block, 8192, rng)
    assert x.shape == (8192, 6) and l.dtype == np.int32 and i.dtype == np.int32
    rows = {tuple(r) for r in np.round(x[:, :3], 5).tolist()}
    assert len(rows) == 8192                                       # enough points: drawn without replacement
    xs, ls, _ = blockio.sample_points(block[:500], 2048, rng)
    assert xs.shape == (2048, 6) and len({tuple(r) for r in xs[:, :3].tolist()}) <= 500   # too few: with replacement
    B = 6
    bi = np.stack([blockio.sample_points(block, 1024, rng)[0] for _ in range(B)])
    bl = np.tile(np.arange(1024, dtype=np.int32), (B, 1))
    bn = np.tile(np.arange(B, dtype=np.int32)[:, None], (1, 1024))       # block id carried in the "inner" slot
    ai, al, an = blockio.augment_batch(bi, bl, bn, np.random.RandomState(4))
    assert ai.shape == bi.shape and ai.dtype == np.float32
    assert sorted(an[:, 0].tolist()) == list(range(B))                    # blocks shuffled, none lost
    assert (al == al[0]).all() and sorted(al[0].tolist()) == list(range(1024))   # one point permutation for all blocks
    for k in range(B):
        src = bi[an[k, 0]][al[k]]
        np.testing.assert_array_equal(ai[k, :, 3:6], src[:, 3:6])          # colours never touched
        d = ai[k, :, 0:3] - src[:, 0:3]
        if k < 2:         # rotated: norms about the origin preserved, z nearly (the random perturbation is <= 0.18 rad)
            np.testing.assert_allclose(np.linalg.norm(ai[k, :, :3], axis=1), np.linalg.norm(src[:, :3], axis=1), rtol=1e-4, atol=1e-5)
            assert np.abs(d).max() > 1e-3
        elif k < 4:       # jittered: clipped at 0.02
            assert 0 < np.abs(d).max() <= 0.02 + 1e-6
        else:
            assert np.abs(d).max() == 0


def test_training_batches_epoch(tmp_path):
    rng = np.random.RandomState(5)
    blocks = _blocks(5, rng)
    paths = []
    for k in range(2):
        p = os.path.join(tmp_path, "f%d.tfrecord" % k)
        sel = blocks[:3] if k == 0 else blocks[3:]
        blockio.write_records(p, [blockio.encode_block(*b) for b in sel])
        paths.append(p)
    got = list(blockio.training_batches(paths, 2, 2048, np.random.RandomState(6), shuffle_buffer=3))
    assert [g[0].shape[0] for g in got] == [2, 2, 1]
    for x, l, i in got:
        assert x.shape[1:] == (2048, 6) and l.shape[1] == 2048 and set(np.unique(i)) <= {0, 1} and l.min() >= 0 and l.max() < 13


# ---- pins against the reference's own numpy (tests/golden/make_blockio_golden.py imports /root/reference in the build
# ---- container; only its inputs / outputs are committed) ---------------------------------------------------------
def _golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "blockio_ref.npz"))


def _same(a, b):
    """bit for bit on this build; a BLAS that fuses the 3-term dot products differently may move a float by one ulp"""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype
    if not np.array_equal(a, b):
        np.testing.assert_array_max_ulp(a, b, maxulp=1)


def test_augmentation_primitives_equal_the_reference_numpy():
    g = _golden()
    xyz = g["xyz"]
    for name in ("rotate_point_cloud", "rotate_perturbation_point_cloud", "jitter_point_cloud"):
        rng = np.random.RandomState(int(g[name + "_seed"]))       # same MT19937 stream as np.random.seed(seed)
        _same(getattr(blockio, name)(xyz.copy(), rng), g[name])


def test_augment_batch_equals_the_reference_augment_fn():
    g = _golden()
    rng = np.random.RandomState(int(g["aug_seed"]))
    x, l, i = blockio.augment_batch(g["aug_in_input"].copy(), g["aug_in_label"].copy(), g["aug_in_inner"].copy(), rng)
    np.testing.assert_array_equal(l, g["aug_out_label"])
    np.testing.assert_array_equal(i, g["aug_out_inner"])
    assert x.dtype == np.float32
    _same(x, g["aug_out_input"].astype(np.float32))             # the float32 feed of the reference's float64 batch
    # the draws really were consumed in the reference's order: the stream is at the same position afterwards
    np.random.seed(int(g["aug_seed"]))
    ref_rng = np.random.RandomState(int(g["aug_seed"]))
    blockio.augment_batch(g["aug_in_input"].copy(), g["aug_in_label"].copy(), g["aug_in_inner"].copy(), ref_rng)
    assert ref_rng.randint(1 << 30) == rng.randint(1 << 30)


def test_block_sampling_rule_equals_the_reference_draws():
    g = _golden()
    rng = np.random.RandomState(int(g["sample_seed"]))
    big = np.arange(1000 * 8, dtype=np.float32).reshape(1000, 8)
    small = np.arange(120 * 8, dtype=np.float32).reshape(120, 8)
    x, _, _ = blockio.sample_points(big, 300, rng)
    np.testing.assert_array_equal(x[:, 0] / 8, g["sample_300_of_1000"])       # row ids: without replacement
    x, _, _ = blockio.sample_points(small, 300, rng)
    np.testing.assert_array_equal(x[:, 0] / 8, g["sample_300_of_120"])        # too few points: with replacement
