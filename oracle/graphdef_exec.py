"""Executes the reference's OWN serialized training graph in numpy.  TEST INFRASTRUCTURE ONLY.

/root/reference/model/air-model.meta is the MetaGraphDef that TensorFlow 1.3 wrote for the
reference's 270k-iteration training run: the exact dataflow that air/air_model.py (+ transformer.py,
vae.py, concrete.py) built -- forward while-loop body, loss, the whole `tf.gradients` backward
(incl. the gradient while-loop, its Stack push/pop pairs, AddN input orders and the single
UnsortedSegmentSum of the sampler), clip_by_global_norm and the 36 ApplyAdam nodes.  TensorFlow is
not installable here, so this module is a small dataflow executor for that graph:

  * a complete protobuf wire-format reader for GraphDef / NodeDef / AttrValue / TensorProto
    (no TF protos needed);
  * one numpy kernel per op type the graph uses (~110), written from TensorFlow 1.3's documented
    op semantics (kernel formulas cited where they matter for rounding: Softplus thresholds,
    LinSpace, AddN / UnsortedSegmentSum accumulation order, ApplyAdam);
  * TF's control-flow semantics for the two while-loop frames (Enter / Merge / Switch /
    NextIteration / Exit / LoopCond), evaluated demand-driven per (frame, iteration), with the
    gradient loop's StackPop resolved to the matching forward-iteration StackPush value, and
    TensorArrays modelled functionally through their flow values.

Variables, the input batch and the graph's (unseeded, seed=seed2=0) Random* nodes are FED.
`float_dtype=np.float64` re-types every float tensor: the same graph evaluated without rounding
noise, which is what pins the *semantics* of the oracle restatement (oracle/air_oracle.py) --
forward, gradients and the Adam update -- to the reference's graph rather than to a reading of its
sources.  tests/test_graph_exec.py holds the assertions; tests/golden/make_graph_golden.py writes
the fixtures the GPU parity tests use.
"""
import struct
import sys
import threading

import numpy as np

# ----------------------------------------------------------------------------- protobuf reader


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


def _signed(v):
    return v if v < (1 << 63) else v - (1 << 64)


def _packed_varints(w, v):
    if w == 0:
        return [_signed(v)]
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(_signed(x))
    return out


_DT = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_}
DT_FLOAT, DT_STRING = 1, 7


def parse_shape(b):
    """TensorShapeProto -> list of ints (-1 unknown) or None (unknown rank)"""
    dims = []
    for f, w, v in fields(b):
        if f == 2:
            size = 0
            for f2, w2, v2 in fields(v):
                if f2 == 1:
                    size = _signed(v2)
            dims.append(size)
        elif f == 3 and v:
            return None
    return dims


def parse_tensor(b):
    """TensorProto -> np.ndarray (strings: object array)"""
    dtype, shape, content, vals = None, [], None, []
    for f, w, v in fields(b):
        if f == 1:
            dtype = v
        elif f == 2:
            shape = parse_shape(v)
        elif f == 4:
            content = v
        elif f == 5:                               # float_val
            vals += [struct.unpack("<f", v)[0]] if w == 5 else list(struct.unpack("<%df" % (len(v) // 4), v))
        elif f == 6:                               # double_val
            vals += [struct.unpack("<d", v)[0]] if w == 1 else list(struct.unpack("<%dd" % (len(v) // 8), v))
        elif f in (7, 10, 11):                     # int_val / int64_val / bool_val
            vals += _packed_varints(w, v)
        elif f == 8:
            vals.append(v)
    if dtype == DT_STRING:
        a = np.empty(len(vals), object)
        a[:] = vals
        return a.reshape(shape) if shape else (a.reshape(()) if len(vals) == 1 else a)
    npdt = _DT[dtype]
    n = int(np.prod(shape)) if shape else 1
    if content is not None:
        return np.frombuffer(content, dtype=npdt).reshape(shape).copy()
    if not vals:
        return np.zeros(shape, npdt)
    a = np.array(vals, dtype=npdt)
    if a.size < n:                                 # TF: the last value fills the remainder
        a = np.concatenate([a, np.full(n - a.size, a[-1], npdt)])
    return a.reshape(shape)


def parse_attr(b):
    """AttrValue -> python value"""
    for f, w, v in fields(b):
        if f == 1:                                 # list(...)
            out = []
            for f2, w2, v2 in fields(v):
                if f2 == 2: out.append(v2.decode("latin1"))
                elif f2 == 3: out += _packed_varints(w2, v2)
                elif f2 == 4: out += [struct.unpack("<f", v2)[0]] if w2 == 5 else list(struct.unpack("<%df" % (len(v2) // 4), v2))
                elif f2 == 5: out += [bool(x) for x in _packed_varints(w2, v2)]
                elif f2 == 6: out += _packed_varints(w2, v2)
                elif f2 == 7: out.append(parse_shape(v2))
                elif f2 == 8: out.append(parse_tensor(v2))
            return out
        if f == 2: return v.decode("latin1")
        if f == 3: return _signed(v)
        if f == 4: return struct.unpack("<f", v)[0]
        if f == 5: return bool(v)
        if f == 6: return v
        if f == 7: return parse_shape(v)
        if f == 8: return parse_tensor(v)
    return None


class Node:
    __slots__ = ("name", "op", "inputs", "ctrl", "attr", "_raw")

    def __init__(self, name, op, inputs, ctrl, raw):
        self.name, self.op, self.inputs, self.ctrl, self._raw, self.attr = name, op, inputs, ctrl, raw, None

    def a(self, key, default=None):
        if self.attr is None:                      # attrs are parsed lazily (Const payloads are big)
            self.attr = {k: parse_attr(v) for k, v in self._raw.items()}
            self._raw = None
        return self.attr.get(key, default)


def _split_input(s):
    if s.startswith("^"):
        return None, s[1:]
    if ":" in s:
        n, i = s.rsplit(":", 1)
        return (n, int(i)), None
    return (s, 0), None


def load_graph(path):
    """MetaGraphDef file -> (tensorflow_version, {name: Node})"""
    raw = open(path, "rb").read()
    graph, version = None, None
    for f, w, v in fields(raw):
        if f == 1:
            for f2, w2, v2 in fields(v):
                if f2 == 5:
                    version = v2.decode()
        elif f == 2:
            graph = v
    nodes = {}
    for f, w, v in fields(graph):
        if f != 1:
            continue
        name = op = None
        ins, ctrl, attrs = [], [], {}
        for f2, w2, v2 in fields(v):
            if f2 == 1: name = v2.decode()
            elif f2 == 2: op = v2.decode()
            elif f2 == 3:
                d, c = _split_input(v2.decode())
                if d is not None: ins.append(d)
                else: ctrl.append(c)
            elif f2 == 5:
                k = val = None
                for f3, w3, v3 in fields(v2):
                    if f3 == 1: k = v3.decode()
                    elif f3 == 2: val = v3
                attrs[k] = val
        nodes[name] = Node(name, op, ins, ctrl, attrs)
    return version, nodes


# ----------------------------------------------------------------------------- op kernels


def _axes(a):
    a = np.asarray(a)
    return tuple(int(x) for x in a.reshape(-1))


def _strided_index(n, begin, end, strides):
    bm, em = n.a("begin_mask", 0), n.a("end_mask", 0)
    elm, nam, sam = n.a("ellipsis_mask", 0), n.a("new_axis_mask", 0), n.a("shrink_axis_mask", 0)
    idx = []
    for i in range(len(begin)):
        bit = 1 << i
        if elm & bit: idx.append(Ellipsis)
        elif nam & bit: idx.append(None)
        elif sam & bit: idx.append(int(begin[i]))
        else:
            idx.append(slice(None if bm & bit else int(begin[i]), None if em & bit else int(end[i]), int(strides[i])))
    return tuple(idx)


def _bcast_grad_args(s0, s1):
    # tensorflow/core/util/bcast.cc: shapes right-aligned and padded with 1; an operand's gradient is
    # summed over every axis where that operand has extent 1 (incl. axes where both have extent 1)
    s0, s1 = [int(x) for x in s0], [int(x) for x in s1]
    n = max(len(s0), len(s1))
    p0, p1 = [1] * (n - len(s0)) + s0, [1] * (n - len(s1)) + s1
    return (np.array([i for i in range(n) if p0[i] == 1], np.int32),
            np.array([i for i in range(n) if p1[i] == 1], np.int32))


def _softplus(x):
    # tensorflow/core/kernels/softplus_op.h (1.3): threshold = log(eps) + 2
    thr = np.log(np.finfo(x.dtype).eps).astype(x.dtype) + x.dtype.type(2)
    ex = np.exp(x)
    return np.where(x > -thr, x, np.where(x < thr, ex, np.log(ex + x.dtype.type(1))))


def _select(c, t, e):
    c = np.asarray(c)
    if c.ndim == 1 and np.ndim(t) > 1:
        c = c.reshape((-1,) + (1,) * (np.ndim(t) - 1))
    return np.where(c, t, e)


def _reduce(fn):
    def k(n, x, ax):
        return fn(x, axis=_axes(ax), keepdims=bool(n.a("keep_dims", False)))
    return k


def _one(x):
    return x.dtype.type(1)


OPS = {
    "Add": lambda n, a, b: a + b, "Sub": lambda n, a, b: a - b, "Mul": lambda n, a, b: a * b,
    "RealDiv": lambda n, a, b: a / b, "Maximum": lambda n, a, b: np.maximum(a, b),
    "Minimum": lambda n, a, b: np.minimum(a, b), "Pow": lambda n, a, b: np.power(a, b),
    "FloorMod": lambda n, a, b: np.mod(a, b), "FloorDiv": lambda n, a, b: np.floor_divide(a, b),
    "Neg": lambda n, a: -a, "Exp": lambda n, a: np.exp(a), "Log": lambda n, a: np.log(a),
    "Sqrt": lambda n, a: np.sqrt(a), "Square": lambda n, a: a * a, "Floor": lambda n, a: np.floor(a),
    "Round": lambda n, a: np.rint(a),                                    # half to even, as TF
    "Reciprocal": lambda n, a: _one(a) / a,
    "Sigmoid": lambda n, a: _one(a) / (_one(a) + np.exp(-a)),          # Eigen scalar_sigmoid_op
    "Tanh": lambda n, a: np.tanh(a), "Relu": lambda n, a: np.maximum(a, a.dtype.type(0)),
    "Softplus": lambda n, a: _softplus(a),
    "SigmoidGrad": lambda n, y, dy: dy * y * (_one(y) - y),
    "TanhGrad": lambda n, y, dy: dy * (_one(y) - y * y),
    "SqrtGrad": lambda n, y, dy: dy * y.dtype.type(0.5) / y,
    "ReluGrad": lambda n, dy, x: dy * (x > 0).astype(dy.dtype),
    "SoftplusGrad": lambda n, dy, x: dy / (np.exp(-x) + _one(x)),
    "BiasAdd": lambda n, x, b: x + b,
    "BiasAddGrad": lambda n, dy: dy.reshape(-1, dy.shape[-1]).sum(axis=0),
    "Less": lambda n, a, b: a < b, "LessEqual": lambda n, a, b: a <= b, "Greater": lambda n, a, b: a > b,
    "GreaterEqual": lambda n, a, b: a >= b, "Equal": lambda n, a, b: a == b,
    "LogicalAnd": lambda n, a, b: np.logical_and(a, b), "LogicalNot": lambda n, a: np.logical_not(a),
    "Select": lambda n, c, t, e: _select(c, t, e),
    "ZerosLike": lambda n, a: np.zeros_like(a),
    "Identity": lambda n, a: a, "StopGradient": lambda n, a: a,
    "Shape": lambda n, a: np.array(np.shape(a), np.int32),
    "ShapeN": lambda n, *xs: tuple(np.array(np.shape(x), np.int32) for x in xs),
    "Rank": lambda n, a: np.array(np.ndim(a), np.int32), "Size": lambda n, a: np.array(np.size(a), np.int32),
    "Reshape": lambda n, a, s: np.reshape(a, _axes(s)),
    "ExpandDims": lambda n, a, d: np.expand_dims(a, int(d)),
    "Squeeze": lambda n, a: np.squeeze(a, axis=tuple(n.a("squeeze_dims") or ()) or None),
    "Transpose": lambda n, a, p: np.transpose(a, _axes(p)),
    "Tile": lambda n, a, m: np.tile(a, _axes(m)),
    "Pack": lambda n, *xs: np.stack(xs, axis=n.a("axis", 0)),
    "Unpack": lambda n, a: tuple(np.moveaxis(a, n.a("axis", 0), 0)),
    "ConcatV2": lambda n, *xs: np.concatenate(xs[:-1], axis=int(xs[-1])),
    "Split": lambda n, d, v: tuple(np.split(v, n.a("num_split"), axis=int(d))),
    "Slice": lambda n, a, b, s: a[tuple(slice(int(b[i]), None if int(s[i]) < 0 else int(b[i]) + int(s[i]))
                                       for i in range(len(b)))],
    "Pad": lambda n, a, p: np.pad(a, [(int(x), int(y)) for x, y in p]),
    "Fill": lambda n, d, v: np.full(_axes(d), v, dtype=np.asarray(v).dtype),
    "Range": lambda n, s, l, d: np.arange(s, l, d, dtype=np.asarray(s).dtype),
    "Gather": lambda n, p, i: p[i],
    "MatMul": lambda n, a, b: np.matmul(a.T if n.a("transpose_a") else a, b.T if n.a("transpose_b") else b),
    "BatchMatMul": lambda n, a, b: np.matmul(np.swapaxes(a, -1, -2) if n.a("adj_x") else a,
                                             np.swapaxes(b, -1, -2) if n.a("adj_y") else b),
    "Sum": _reduce(np.sum), "Prod": _reduce(np.prod), "Any": _reduce(np.any), "Mean": _reduce(np.mean),
    "L2Loss": lambda n, a: np.sum(a * a) / a.dtype.type(2),
    "BroadcastGradientArgs": lambda n, a, b: _bcast_grad_args(a, b),
    "NoOp": lambda n, *a: None, "ControlTrigger": lambda n, *a: None,
}


def _k_linspace(n, start, stop, num):
    # tensorflow/core/kernels/sequence_ops.cc: flat(i) = start + step * i in T
    start, stop, num = np.asarray(start), np.asarray(stop), int(num)
    T = start.dtype.type
    if num == 1:
        return np.array([start], start.dtype)
    step = (T(stop) - T(start)) / T(num - 1)
    return np.array([T(start) + step * T(i) for i in range(num)], start.dtype)


def _k_addn(n, *xs):
    acc = xs[0]                                     # aggregate_ops: In0 + In1 + ... left to right
    for x in xs[1:]:
        acc = acc + x
    return acc


# "sequential": TF's CPU kernel.  "carried16": what backward="reference_carried" computes -- the graph's one
# UnsortedSegmentSum takes the four Gather gradients concatenated (taps a, b, c, d); segments with short streams are summed
# as the CPU kernel sums them, the long ones through oracle.air_oracle.carried_segment_sum.
# A module switch for tests / tests/golden/make_graph_golden.py:
# the whole saved graph executed with that ONE kernel swapped shows what the order does to the 36 gradients.
SEGMENT_SUM_ORDER = "sequential"


def _k_unsorted_segment_sum(n, data, ids, num):
    ids = np.asarray(ids)
    out = np.zeros((int(num),) + data.shape[ids.ndim:], data.dtype)
    flat_ids, flat = ids.reshape(-1), data.reshape((-1,) + data.shape[ids.ndim:])
    if SEGMENT_SUM_ORDER == "carried16":
        # backward="reference_carried": short streams as the CPU kernel, long ones through oracle.air_oracle.carried_segment_sum
        from oracle.air_oracle import carried_segment_sum
        assert int(np.prod(flat.shape[1:])) == 1 and flat.shape[0] % 4 == 0, "the sampler's scatter: four concatenated tap gradients"
        flat, q = flat.reshape(-1), flat.shape[0] // 4
        return carried_segment_sum([flat_ids[k * q:(k + 1) * q] for k in range(4)], [flat[k * q:(k + 1) * q] for k in range(4)],
                                   int(num)).reshape(out.shape)
    else:
        assert SEGMENT_SUM_ORDER == "sequential", SEGMENT_SUM_ORDER
    np.add.at(out, flat_ids, flat)                                                 # in index order, like the CPU kernel
    return out


def _k_strided_slice(n, a, b, e, s):
    return a[_strided_index(n, b, e, s)]


def _k_strided_slice_grad(n, shape, b, e, s, dy):
    out = np.zeros(_axes(shape), dy.dtype)
    out[_strided_index(n, b, e, s)] = dy
    return out


def _k_concat_offset(n, dim, *shapes):
    dim, off, outs = int(dim), 0, []
    for s in shapes:
        o = np.zeros(len(s), np.int32)
        o[dim] = off
        off += int(s[dim])
        outs.append(o)
    return tuple(outs)


def _k_dynamic_stitch(n, *xs):
    N = n.a("N")
    idx, data = xs[:N], xs[N:]
    size = max(int(np.max(i)) for i in idx if np.size(i)) + 1
    tail = None
    for i, d in zip(idx, data):
        if np.size(i):
            tail = np.shape(d)[np.ndim(i):]
    out = np.zeros((size,) + tuple(tail), np.asarray(data[0]).dtype)
    for i, d in zip(idx, data):
        i = np.asarray(i).reshape(-1)
        d = np.asarray(d).reshape((-1,) + tuple(tail))
        for j in range(len(i)):
            out[i[j]] = d[j]
    return out


def apply_adam(var, m, v, b1p, b2p, lr, b1, b2, eps, g):
    """tensorflow/core/kernels/training_ops.cc ApplyAdam (use_nesterov = False):
         alpha = lr * sqrt(1 - beta2_power) / (1 - beta1_power)
         m += (g - m) * (1 - beta1); v += (g*g - v) * (1 - beta2); var -= (m * alpha) / (sqrt(v) + eps)
    Returns (var, m, v) after the update."""
    T = var.dtype.type
    alpha = lr * np.sqrt(T(1) - b2p) / (T(1) - b1p)
    m2 = m + (g - m) * (T(1) - b1)
    v2 = v + (g * g - v) * (T(1) - b2)
    return var - (m2 * alpha) / (np.sqrt(v2) + eps), m2, v2


OPS.update({"LinSpace": _k_linspace, "AddN": _k_addn, "UnsortedSegmentSum": _k_unsorted_segment_sum,
            "StridedSlice": _k_strided_slice, "StridedSliceGrad": _k_strided_slice_grad,
            "ConcatOffset": _k_concat_offset, "DynamicStitch": _k_dynamic_stitch})

_CONTROL = {"Enter", "RefEnter", "Merge", "Switch", "NextIteration", "Exit", "LoopCond"}


class TensorArrayFlow(dict):
    """functional model of a TensorArray's contents, carried by its `flow` value"""


class Executor:
    """Demand-driven evaluation of tensors of the graph.

    feeds: {tensor name ("node" or "node:i"): value | callable(iteration) for nodes inside a loop}.
    Every VariableV2 that is reached, the dequeue node and the Random* nodes of the loop body must be fed."""

    def __init__(self, nodes, feeds, float_dtype=np.float32):
        self.nodes, self.fd = nodes, np.dtype(float_dtype)
        self.feeds = {}
        for k, v in feeds.items():
            self.feeds[k if ":" in k else k + ":0"] = v
        self.memo = {}
        self.trips = {}
        self.assigned = {}                     # ApplyAdam / AssignAdd results: var name -> dict
        self._frames()

    # ---- frame membership: a node lives in the frame of the Enter nodes it (transitively) consumes
    def _frames(self):
        nodes = self.nodes
        cons = {}
        for n in nodes.values():
            for (src, _i) in n.inputs:
                cons.setdefault(src, []).append(n.name)
            for src in n.ctrl:
                cons.setdefault(src, []).append(n.name)
        frame, work = {}, []
        self.loopcond, self.exits = {}, {}
        for n in nodes.values():
            if n.op in ("Enter", "RefEnter"):
                frame[n.name] = n.a("frame_name")
                work.append(n.name)
        while work:
            cur = work.pop()
            f = frame[cur]
            if nodes[cur].op == "Exit":
                continue
            for c in cons.get(cur, ()):
                if nodes[c].op in ("Enter", "RefEnter"):
                    continue
                if c in frame:
                    if frame[c] != f:
                        raise ValueError("node %s reached from frames %s and %s" % (c, frame[c], f))
                    continue
                frame[c] = f
                work.append(c)
        for name, f in frame.items():
            if nodes[name].op == "LoopCond":
                self.loopcond[f] = name
        self.frame = frame

    # ---- evaluation
    def _cast_float(self, a):
        if isinstance(a, np.ndarray) and a.dtype == np.float32 and self.fd != np.float32:
            return a.astype(self.fd)
        return a

    def run(self, fetches, iters=None):
        """evaluates the fetches ("node" / "node:i") in a big-stack thread.  Tensors inside a
        while-loop frame need `iters` = {frame_name: iteration} (FWD_FRAME / BWD_FRAME below)."""
        out, err = {}, []
        iters = dict(iters or {})

        def work():
            try:
                for t in fetches:
                    name, idx = _split_input(t)[0]
                    out[t] = self.ev(name, idx, iters)
            except BaseException as e:              # noqa: BLE001 -- re-raised in the caller's thread
                err.append(e)
        old = sys.getrecursionlimit()
        sys.setrecursionlimit(1000000)
        threading.stack_size(1024 * 1024 * 1024)
        th = threading.Thread(target=work)
        th.start()
        th.join()
        threading.stack_size(0)
        sys.setrecursionlimit(old)
        if err:
            raise err[0]
        return [out[t] for t in fetches]

    def trip_count(self, f):
        """number of iterations of frame f whose LoopCond was true"""
        if f not in self.trips:
            i = 0
            while bool(self.ev(self.loopcond[f], 0, {f: i})):
                i += 1
                if i > 10000:
                    raise RuntimeError("loop %s does not terminate" % f)
            self.trips[f] = i
        return self.trips[f]

    def ev(self, name, idx, it):
        f = self.frame.get(name)
        if self.nodes[name].op == "Exit":
            f = None
        key = (name, it[f]) if f is not None else name
        r = self.memo.get(key)
        if r is None:
            fk = "%s:%d" % (name, idx)
            if fk in self.feeds:
                v = self.feeds[fk]
                return v(it[f]) if callable(v) else v
            r = self._eval(self.nodes[name], f, it)
            if not isinstance(r, tuple):
                r = (r,)
            self.memo[key] = r
        return r[idx]

    def _in(self, n, k, it):
        s, i = n.inputs[k]
        return self.ev(s, i, it)

    def _stack_of(self, n):
        """Stack node behind the handle input of a StackPush / StackPop"""
        src = self.nodes[n.inputs[0][0]]
        while src.op in ("RefEnter", "Enter", "Identity"):
            src = self.nodes[src.inputs[0][0]]
        assert src.op == "Stack", src.op
        return src.name

    def _eval(self, n, f, it):
        op = n.op
        if op == "Const":
            return self._cast_float(n.a("value"))
        if op in ("Enter", "RefEnter"):
            outer = {k: v for k, v in it.items() if k != f}
            return self._in(n, 0, outer)
        if op == "Merge":
            srcs = [self.nodes[s] for s, _ in n.inputs]
            ent = [k for k, s in enumerate(srcs) if s.op in ("Enter", "RefEnter")]
            nxt = [k for k, s in enumerate(srcs) if s.op == "NextIteration"]
            assert len(ent) == 1 and len(nxt) == 1, (n.name, [s.op for s in srcs])
            if it[f] == 0:
                return (self._in(n, ent[0], it), np.int32(ent[0]))
            prev = dict(it); prev[f] = it[f] - 1
            return (self._in(srcs[nxt[0]], 0, prev), np.int32(nxt[0]))
        if op == "Switch":
            v = self._in(n, 0, it)          # loop Switch: :1 while the condition holds, :0 on exit
            return (v, v)
        if op in ("NextIteration", "LoopCond"):
            return self._in(n, 0, it)
        if op == "Exit":
            fr = self.frame[n.name]
            last = dict(it); last[fr] = self.trip_count(fr)
            return self._in(n, 0, last)
        if op == "Stack":
            return n.name
        if op == "StackPush":
            return self._in(n, 1, it)
        if op == "StackPop":
            st = self._stack_of(n)
            push = self._pushes()[st]
            ff = self.frame[push.name]
            fwd = {ff: self.trip_count(ff) - 1 - it[f]}
            return self._in(push, 1, fwd)
        if op == "TensorArrayV3":
            return (n.name, TensorArrayFlow())
        if op == "TensorArrayWriteV3":
            flow = TensorArrayFlow(self._in(n, 3, it))
            flow[int(self._in(n, 1, it))] = self._in(n, 2, it)
            return flow
        if op == "TensorArraySizeV3":
            return np.int32(len(self._in(n, 1, it)))
        if op == "TensorArrayGatherV3":
            flow = self._in(n, 2, it)
            return np.stack([flow[int(i)] for i in self._in(n, 1, it)])
        if op == "Cast":
            x = self._in(n, 0, it)
            dst = n.a("DstT")
            return np.asarray(x).astype(self.fd if dst == DT_FLOAT else _DT[dst])
        if op == "ApplyAdam":
            var, m, v, b1p, b2p, lr, b1, b2, eps, g = (self._in(n, k, it) for k in range(10))
            new, m2, v2 = apply_adam(var, m, v, b1p, b2p, lr, b1, b2, eps, g)
            self.assigned[n.inputs[0][0]] = dict(var=new, m=m2, v=v2, grad=g)
            return new
        if op == "AssignAdd":
            new = self._in(n, 0, it) + self._in(n, 1, it)
            self.assigned[n.inputs[0][0]] = dict(var=new)
            return new
        if op == "VariableV2":
            raise KeyError("variable %s is not fed" % n.name)
        if op in ("RandomUniform", "RandomStandardNormal", "QueueDequeueManyV2", "Placeholder"):
            raise KeyError("%s node %s must be fed" % (op, n.name))
        k = OPS.get(op)
        if k is None:
            raise NotImplementedError("op %s (node %s)" % (op, n.name))
        args = [self.ev(s, i, it) for s, i in n.inputs]
        return k(n, *args)

    def _pushes(self):
        if not hasattr(self, "_push_map"):
            self._push_map = {}
            for n in self.nodes.values():
                if n.op == "StackPush":
                    st = self._stack_of(n)
                    assert st not in self._push_map, "two pushes onto %s" % st
                    self._push_map[st] = n
        return self._push_map


# ----------------------------------------------------------------------------- AIR-specific driver

SCOPE = "air/rnn/"
W = "air/rnn/while/"
G = "air/training/gradients/"
FWD_FRAME = "air/rnn/while/air/rnn/while/"
BWD_FRAME = "air/training/gradients/air/rnn/while/air/rnn/while/"
# the five unseeded RNG nodes of the loop body (air_model.py:127 x2, vae.py:23, 37, concrete.py:23)
RNG_NODES = {
    "eps_scale": W + "scale/random_normal/RandomStandardNormal",
    "eps_shift": W + "shift/random_normal/RandomStandardNormal",
    "eps_z": W + "vae/rec_sample/random_normal/RandomStandardNormal",
    "eps_x": W + "vae/gen_sample/random_normal/RandomStandardNormal",
    "u": W + "z_pres/gumbel/random_uniform/RandomUniform",
}


def air_feeds(nodes, params, images, targets, noise, global_step=0, adam=None, float_dtype=np.float32):
    """feeds for the train model `air/`: variables (TF names relative to air/rnn/, as in
    oracle.air_oracle.param_shapes), the dequeued batch, the injected noise [N,B,...] per loop
    iteration, global_step and (optionally) the Adam slots {name: (m, v)} / beta powers."""
    fd = np.dtype(float_dtype)
    feeds = {"pipeline/shuffle_batch:0": np.asarray(images, fd), "pipeline/shuffle_batch:1": np.asarray(targets, np.int64),
             "air/global_step": np.int32(global_step)}
    for k, v in params.items():
        feeds[SCOPE + k] = np.asarray(v, fd)
    for k, node in RNG_NODES.items():
        arr = np.asarray(noise[k], fd)
        feeds[node] = (lambda i, arr=arr: arr[i])
    # the beta power accumulators after `global_step` updates: initialised to beta and multiplied by
    # the (fp32) beta constant once per apply_gradients (tf.train.AdamOptimizer._finish)
    b1p, b2p = np.float32(0.9), np.float32(0.999)
    for _ in range(global_step):
        b1p, b2p = b1p * np.float32(0.9), b2p * np.float32(0.999)
    feeds["air/training/beta1_power"] = fd.type(b1p)
    feeds["air/training/beta2_power"] = fd.type(b2p)
    for k, v in params.items():
        m, vv = (adam[k] if adam else (np.zeros_like(v), np.zeros_like(v)))
        feeds["air/training/" + SCOPE + k + "/Adam"] = np.asarray(m, fd)
        feeds["air/training/" + SCOPE + k + "/Adam_1"] = np.asarray(vv, fd)
    return feeds


def test_model_feeds(params, images, targets, noise, global_step=0, float_dtype=np.float32):
    """feeds for the test model `air_1/` (reuse=True, train=False, dynamic batch: training.py:109-123)"""
    fd = np.dtype(float_dtype)
    feeds = {"pipeline/Placeholder": np.asarray(images, fd), "pipeline/Placeholder_1": np.asarray(targets, np.int32),
             "air/global_step": np.int32(global_step)}
    for k, v in params.items():
        feeds[SCOPE + k] = np.asarray(v, fd)
    for k, node in RNG_NODES.items():
        arr = np.asarray(noise[k], fd)
        feeds[node.replace("air/", "air_1/", 1)] = (lambda i, arr=arr: arr[i])
    return feeds


# The attributes the reference's callers fetch (training.py:212-224, demo/model_wrapper.py:22-25,
# air_model.py:568-611) -> the tensor of the SAVED graph that holds them.  The graph was saved from a
# slightly older source revision: loss / accuracy means live under summaries/, and `rec_windows_1`
# (the latents array) holds the SAMPLE where air/vae.py:43 now returns the mean.
def output_tensors(scope="air"):
    s = scope + "/"
    return {
        "loss": s + "summaries/loss", "accuracy": s + "summaries/accuracy",
        "reconstruction": s + "loss/reconstruction/clipped_rec",
        "reconstruction_loss": s + "loss/reconstruction/Neg",
        "rec_num_digits": s + "rnn/while/Exit_6",
        "rec_scales": s + "rec_scales", "rec_shifts": s + "rec_shifts", "rec_st_back": s + "rec_st_back",
        "rec_windows": s + "rec_windows", "_graph_latent_samples": s + "rec_windows_1",
        "z_pres_probs": s + "z_pres_probs", "z_pres_kls": s + "z_pres_kls", "scale_kls": s + "scale_kls",
        "shift_kls": s + "shift_kls", "vae_kls": s + "vae_kls",
        "z_pres_prior_log_odds": s + "z_pres_prior_log_odds_log",
    }


def adam_nodes(nodes):
    """{variable name (relative to air/rnn/): ApplyAdam node}"""
    return {n.inputs[0][0][len(SCOPE):]: n for n in nodes.values()
            if n.op == "ApplyAdam" and n.name.startswith("air/")}


def raw_gradient_tensor(nodes, adam_node):
    """tensor name of the UNCLIPPED gradient of the variable an ApplyAdam node updates: its grad
    input is clip_by_global_norm's `grad * scale` (air_model.py:673)"""
    t = adam_node.inputs[9]
    x = nodes[t[0]]
    while x.op == "Identity":
        t = x.inputs[0]
        x = nodes[t[0]]
    assert x.op == "Mul", x.op
    return "%s:%d" % x.inputs[0]


# tensors of the gradient loop body that are the interface of the two sampler backward kernels
# (backward-frame iteration j <-> forward step T' - 1 - j)
SAMPLER_BWD_TENSORS = {
    "g_select": G + W + "canvas/add_grad/tuple/control_dependency_1",        # d loss / d (masked z*window_recon) [B,D]
    "g_window_recon": G + W + "canvas/Reshape_grad/Reshape",                 # z * that, [B,C,C]
    "d_z_canvas": G + W + "canvas/ExpandDims_grad/Reshape",                  # [B]
    "d_vae_recon": G + W + "st_backward/Reshape_grad/Reshape",               # [B,d]: the UnsortedSegmentSum result
    "d_theta_recon": G + W + "st_backward/SpatialTransformer/_transform/Reshape_grad/Reshape",   # [B,2,3]
    "d_window": G + W + "st_forward/strided_slice_grad/StridedSliceGrad",    # [B,w,w,1] d loss / d glimpse
    "d_theta": G + W + "st_forward/SpatialTransformer/_transform/Reshape_grad/Reshape",          # [B,2,3]
    "d_scale": G + "AddN_24", "d_shift_x": G + "AddN_23", "d_shift_y": G + "AddN_25",           # [B] totals wrt s, x, y
    "d_gen_pre": G + W + "vae/gen_sample/Sigmoid_grad/SigmoidGrad",          # [B,d] grad wrt the decoder's pre-sigmoid output
    "d_s_write_0": G + W + "st_backward/truediv_grad/tuple/control_dependency_1",     # the write's four legs into s
    "d_s_write_1": G + W + "st_backward/truediv_1_grad/tuple/control_dependency_1",
    "d_s_write_2": G + W + "st_backward/truediv_2_grad/tuple/control_dependency_1",
    "d_s_write_3": G + W + "st_backward/truediv_3_grad/tuple/control_dependency_1",
    "d_x_write": G + W + "st_backward/Neg_grad/Neg", "d_y_write": G + W + "st_backward/Neg_1_grad/Neg",
    # gradients wrt the five head outputs (inputs of their BiasAddGrad nodes)
    "d_out_scale_mean": G + "AddN_27", "d_out_scale_lv": G + "AddN_31", "d_out_shift_mean": G + "AddN_28",
    "d_out_shift_lv": G + "AddN_32",
    "d_out_z_log_odds": G + W + "z_pres/log_odds/output/strided_slice_grad/StridedSliceGrad",
}
# forward tensors of the loop body that the sampler kernels take as inputs
SAMPLER_FWD_TENSORS = {
    "z_pres": W + "z_pres/gumbel/Sigmoid", "z_pre": W + "z_pres/gumbel/truediv", "mask": W + "canvas/Less",
    "mask_prev": W + "loss/z_pres_kl/Less", "vae_recon": W + "vae/gen_sample/Sigmoid",
    "s": W + "scale/strided_slice", "x": W + "shift/strided_slice", "y": W + "shift/strided_slice_1",
    "out_scale_mean": W + "scale/mean/output/BiasAdd", "out_scale_lv": W + "scale/log_variance/output/BiasAdd",
    "out_shift_mean": W + "shift/mean/output/BiasAdd", "out_shift_lv": W + "shift/log_variance/output/BiasAdd",
    "out_z_log_odds": W + "z_pres/log_odds/output/BiasAdd",
}
