"""Generates tests/golden/tf_index_listing.json: the entries of the reference's own checkpoint
index (/root/reference/model/air-model.index, written by TensorFlow 1.3 at step 270 000) as parsed
by tf_checkpoint.read_index -- names, dtypes, shapes, offsets, sizes and TensorFlow's masked
CRC-32C values.  Data only; run in the build container: python tests/golden/make_tf_index_listing.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import tf_checkpoint as tfc  # noqa: E402

header, entries = tfc.read_index("/root/reference/model/air-model.index", verify=True)
out = {"header": header, "file_bytes": os.path.getsize("/root/reference/model/air-model.index"),
       "entries": {k: v for k, v in entries.items()}}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf_index_listing.json"), "w"), indent=0)
print(len(entries), "entries,", sum(v["size"] for v in entries.values()), "data bytes")
