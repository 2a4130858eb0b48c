"""Reads the reference's serialized training graph (model/air-model.meta, a MetaGraphDef
written by TF 1.3 after 270k iterations) with a plain protobuf wire-format walker -- no
TensorFlow needed -- and extracts the facts the oracle restatement relies on but the Python
sources do not show (LSTM gate order / forget bias, FC activations, initializer limits, Adam
and clip constants, RNG seeds, annealing constants).  TEST INFRASTRUCTURE ONLY.

  python oracle/graphdef_pin.py /root/reference/model/air-model.meta > tests/golden/graphdef_facts.json
"""
import json
import struct
import sys
from collections import Counter


def _varint(b, i):
    r, s = 0, 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if not c & 0x80:
            return r, i
        s += 7


def fields(b):
    """yields (field_no, wire_type, value) of one message"""
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(b, i)
        elif w == 1:
            v = b[i:i + 8]; i += 8
        elif w == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]; i += ln
        elif w == 5:
            v = b[i:i + 4]; i += 4
        else:
            raise ValueError("wire type %d" % w)
        yield f, w, v


def parse_attr(b):
    """AttrValue: 2 s, 3 i, 4 f, 5 b, 6 type, 7 shape, 8 tensor, 1 list"""
    out = {}
    for f, w, v in fields(b):
        if f == 2: out["s"] = v.decode("latin1")
        elif f == 3: out["i"] = v if v < (1 << 63) else v - (1 << 64)
        elif f == 4: out["f"] = struct.unpack("<f", v)[0]
        elif f == 5: out["b"] = bool(v)
        elif f == 6: out["type"] = v
        elif f == 8: out["tensor"] = parse_tensor(v)
    return out


def parse_tensor(b):
    """TensorProto: 1 dtype, 2 shape, 4 tensor_content, 5 float_val, 7 int_val"""
    t = {"dtype": None, "shape": [], "vals": []}
    for f, w, v in fields(b):
        if f == 1: t["dtype"] = v
        elif f == 2:
            for f2, w2, v2 in fields(v):
                if f2 == 2:
                    for f3, w3, v3 in fields(v2):
                        if f3 == 1: t["shape"].append(v3)
        elif f == 4:
            if t["dtype"] == 1: t["vals"] = list(struct.unpack("<%df" % (len(v) // 4), v))[:16]
            elif t["dtype"] == 3: t["vals"] = list(struct.unpack("<%di" % (len(v) // 4), v))[:16]
        elif f == 5:
            if w == 5: t["vals"].append(struct.unpack("<f", v)[0])
            else: t["vals"] += list(struct.unpack("<%df" % (len(v) // 4), v))
        elif f == 7:
            if w == 0: t["vals"].append(v)
    return t


def load_nodes(path):
    raw = open(path, "rb").read()
    graph = None
    version = None
    for f, w, v in fields(raw):
        if f == 1:
            for f2, w2, v2 in fields(v):
                if f2 == 5: version = v2.decode()
        if f == 2: graph = v
    nodes = {}
    for f, w, v in fields(graph):
        if f != 1: continue
        nd = {"inputs": [], "attr": {}}
        for f2, w2, v2 in fields(v):
            if f2 == 1: nd["name"] = v2.decode()
            elif f2 == 2: nd["op"] = v2.decode()
            elif f2 == 3: nd["inputs"].append(v2.decode())
            elif f2 == 5:
                k = val = None
                for f3, w3, v3 in fields(v2):
                    if f3 == 1: k = v3.decode()
                    elif f3 == 2: val = parse_attr(v3)
                nd["attr"][k] = val
        nodes[nd["name"]] = nd
    return version, nodes


def const(nodes, name):
    t = nodes[name]["attr"]["value"]["tensor"]
    return t["vals"][0] if len(t["vals"]) == 1 else t["vals"]


def main(path):
    version, nodes = load_nodes(path)
    W = "air/rnn/while/"
    facts = {"tensorflow_version": version, "num_nodes": len(nodes)}
    body = {n: d for n, d in nodes.items() if n.startswith(W)}
    facts["body_nodes_non_const"] = sum(1 for d in body.values() if d["op"] != "Const")
    facts["op_histogram_top"] = dict(Counter(d["op"] for d in nodes.values()).most_common(25))
    # LSTM: split order and forget bias
    facts["lstm_split_num"] = nodes[W + "rnn/split"]["attr"]["num_split"]["i"]
    # the cell's pointwise wiring: which Split output feeds which nonlinearity (gate order i, j, f, o)
    strip = lambda i: i[len(W):] if i.startswith(W) else i
    facts["lstm_cell_wiring"] = {n[len(W):]: [d["op"]] + [strip(i) for i in d["inputs"]]
                                 for n, d in body.items()
                                 if n.startswith(W + "rnn/") and not n.startswith(W + "rnn/rnn_1/")
                                 and d["op"] != "Const" and "Enter" not in n}
    lstm_nodes = sorted(n for n in body if n.startswith(W + "rnn/rnn_1/") and body[n]["op"] != "Const")
    facts["lstm_ops"] = {n[len(W):]: [body[n]["op"]] + body[n]["inputs"] for n in lstm_nodes if "Enter" not in n}
    for n in body:
        if n.startswith(W + "rnn/rnn_1/") and body[n]["op"] == "Const":
            facts.setdefault("lstm_consts", {})[n[len(W):]] = const(nodes, n)
    # activations of every fully connected layer in the body
    acts = {}
    for n, d in body.items():
        if d["op"] in ("Relu", "Softplus", "Sigmoid", "Tanh", "Exp", "Sqrt", "Round", "Log", "Floor"):
            acts[n[len(W):]] = d["op"]
    facts["body_activations"] = dict(sorted(acts.items()))
    # all scalar constants in the body (eps, thresholds, prior params ...)
    sc = {}
    for n, d in body.items():
        if d["op"] == "Const":
            t = d["attr"]["value"]["tensor"]
            if len(t["vals"]) == 1 and not t["shape"]:
                sc[n[len(W):]] = t["vals"][0]
    facts["body_scalar_consts"] = dict(sorted(sc.items()))
    # RNG ops and seeds
    facts["rng_ops"] = {n: [d["op"], d["attr"].get("seed", {}).get("i"), d["attr"].get("seed2", {}).get("i")]
                        for n, d in nodes.items() if d["op"] in ("RandomStandardNormal", "RandomUniform") and n.startswith(W)}
    # initializer limits
    init = {}
    for n, d in nodes.items():
        if n.startswith("air/") and n.endswith("Initializer/random_uniform/max"):
            init[n.replace("/Initializer/random_uniform/max", "")] = const(nodes, n)
    facts["initializer_uniform_max"] = dict(sorted(init.items()))
    # optimizer / clipping / annealing constants
    tr = {}
    for n, d in nodes.items():
        if d["op"] == "Const" and (n.startswith("air/training/") or n.startswith("air/z_pres_prior_log_odds")) \
                and "gradients" not in n and "Initializer" not in n:
            t = d["attr"]["value"]["tensor"]
            if len(t["vals"]) == 1:
                tr[n] = t["vals"][0]
    facts["training_scalar_consts"] = dict(sorted(tr.items()))
    facts["apply_adam_count"] = sum(1 for d in nodes.values() if d["op"] == "ApplyAdam")
    facts["unsorted_segment_sum_count"] = sum(1 for d in nodes.values() if d["op"] == "UnsortedSegmentSum")
    json.dump(facts, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/model/air-model.meta")
