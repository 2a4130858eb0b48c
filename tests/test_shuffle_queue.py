"""The training driver's input queue (air_shuffle_batch_*, csrc/air_input.hip) against a numpy model of the reference's
tf.train.shuffle_batch (oracle/shuffle_queue.py; /root/reference/multi_mnist.py:240-249, training.py:76-81).
CPU: what the model's batches look like (the statistics the reference's queue has: a record leaves the 10 640-slot window
after a geometric waiting time, batches mix neighbouring epochs, nothing is lost or duplicated).  GPU: the kernel's
picks equal the model's pick for pick, under stream capture too."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import shuffle_queue as sq

N_REC, CAP, BATCH, MIN_AFTER = 60000, 10640, 64, 10000       # multi_mnist.py defaults: 60 000 train canvases; :246-247


def test_philox_known_answer():
    # Random123 kat_vectors: philox4x32-10, counter = key = 0 / all ones / pi digits
    assert sq.philox4x32_10((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert sq.philox4x32_10((0xffffffff,) * 4, (0xffffffff, 0xffffffff)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert sq.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_queue_model_statistics_are_the_shuffle_queues():
    nb = 3000                                                # 192 000 records: 3.2 epochs
    b = sq.tf_queue_batches(N_REC, CAP, BATCH, MIN_AFTER, nb, lambda n, k: sq.device_draws(5, n, k))
    assert b.shape == (nb, BATCH) and b.min() >= 0 and b.max() < N_REC
    # nothing lost, nothing duplicated: what came out + what is still resident = the stream so far
    flat = b.reshape(-1)
    counts = np.bincount(flat, minlength=N_REC)
    enq = CAP + (nb - 1) * BATCH                            # records enqueued before the last batch was taken
    full, part = divmod(enq, N_REC)
    upper = full + (np.arange(N_REC) < part)
    assert (counts <= upper).all() and upper.sum() - counts.sum() == CAP - BATCH
    # a batch is a uniform draw WITHOUT replacement from the resident window: no repeats inside a batch early on
    assert all(len(set(row)) == BATCH for row in b[:100])
    # waiting time: a record enqueued at stream position p leaves with probability ~ batch / capacity per step --
    # mean residence ~ capacity / batch = 166 batches; the window lags the stream head by up to ~1000+ batches for a few
    head = CAP + np.arange(nb)[:, None] * BATCH              # stream position when batch n is taken
    first_epoch = b[: (N_REC - CAP) // BATCH]                # while positions == record ids
    lag = (head[: len(first_epoch)] - first_epoch) / BATCH   # in batches
    assert lag.min() >= 0
    steady = lag[400:]                                       # (the initial fill is all "young": a transient of a few residences)
    assert 150 < steady.mean() < 185, steady.mean()
    assert np.percentile(steady, 99) > 600
    # epochs mix: around an epoch boundary a batch holds records of both epochs
    n_b = (N_REC - CAP) // BATCH + 20                        # shortly after the stream wrapped
    row = b[n_b]
    assert (row < 5000).any() and (row > 40000).any()


def test_parallel_pick_resolution_is_the_sequential_queue():
    """the rule the device kernel uses (every pick resolved on its own by walking the earlier picks of the batch backwards,
    oracle.shuffle_queue.parallel_dequeue) against the literal pop-and-swap queue: tiny capacities and draws that collide
    all the time, batches that consume most of the queue, the reference's geometry"""
    rng = np.random.RandomState(0)
    for trial in range(400):
        batch = int(rng.choice([4, 8, 16, 64]))
        cap = batch + int(rng.randint(0, 3 * batch)) + (0 if rng.rand() < 0.5 else 200)
        n_rec = 997
        q = [int(v) for v in rng.permutation(cap + 1000)[:cap]]
        pos = int(rng.randint(0, 1000))
        for _ in range(3):
            r = [int(v) for v in rng.randint(0, 2 ** 32, size=batch, dtype=np.uint64)]
            if rng.rand() < 0.3:
                r = [int(v) for v in rng.randint(0, 5, size=batch)]         # nearly every pick collides with an earlier one
            lit, size, picks = list(q), cap, []
            for k in range(batch):                                         # RandomShuffleQueue::TryDequeueMany, literally
                i = r[k] % size
                picks.append(lit[i])
                lit[i] = lit[size - 1]
                size -= 1
            for t in range(batch):
                lit[cap - batch + t] = (pos + t) % n_rec
            got, new = sq.parallel_dequeue(q, batch, r, pos, n_rec)
            assert got == picks and new == lit
            q, pos = lit, pos + batch
    # the reference's geometry, against the batch-level model
    seed, nb = 11, 30
    ref = sq.tf_queue_batches(N_REC, CAP, BATCH, MIN_AFTER, nb, lambda n, k: sq.device_draws(seed, n, k))
    q, pos = [i % N_REC for i in range(CAP)], CAP
    for n in range(nb):
        got, q = sq.parallel_dequeue(q, BATCH, sq.device_draws(seed, n, BATCH), pos, N_REC)
        pos += BATCH
        assert got == list(ref[n])


needs_gpu = pytest.mark.gpu


def _struct(H, q, st, picks, seed, cap=CAP, batch=BATCH, mad=MIN_AFTER, n=N_REC):
    return H.ShuffleBatch(q.data_ptr(), st.data_ptr(), picks.data_ptr(), cap, batch, mad, n, seed)


def test_argument_errors_without_gpu():
    from air import _hip as H
    lib = H.lib()
    assert lib.air_shuffle_batch_dequeue(None, None) == -1
    buf = (C.c_int64 * 8)()
    A = C.addressof(buf)
    mk = lambda **kw: H.ShuffleBatch(**dict(dict(queue=A, state=A, picks=A, capacity=CAP, batch=BATCH, min_after_dequeue=MIN_AFTER,  # noqa: E731
                                                 n_records=N_REC, seed=1), **kw))
    assert lib.air_shuffle_batch_init(C.byref(mk(batch=62)), None) == -3            # a multiple of 4
    assert lib.air_shuffle_batch_init(C.byref(mk(capacity=10000)), None) == -1      # capacity - batch < min_after_dequeue
    assert lib.air_shuffle_batch_init(C.byref(mk(capacity=20000)), None) == -2      # does not fit the LDS image
    assert lib.air_shuffle_batch_init(C.byref(mk(queue=None)), None) == -1


@needs_gpu
@pytest.mark.parametrize("cap,batch,mad,n_rec", [(CAP, BATCH, MIN_AFTER, N_REC), (100, 8, 50, 37), (640, 64, 0, 1000)])
def test_kernel_equals_the_queue_model_pick_for_pick(cap, batch, mad, n_rec):
    from air import _hip as H
    nb, seed = 400, 0x1234567890ab
    q = torch.zeros(cap, dtype=torch.int32, device="cuda")
    st = torch.zeros(2, dtype=torch.int64, device="cuda")
    picks = torch.zeros(batch, dtype=torch.int32, device="cuda")
    a = _struct(H, q, st, picks, seed, cap, batch, mad, n_rec)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    H.check(H.lib().air_shuffle_batch_init(C.byref(a), s))
    got = []
    for _ in range(nb):
        H.check(H.lib().air_shuffle_batch_dequeue(C.byref(a), s))
        got.append(picks.clone())
    torch.cuda.synchronize()
    got = torch.stack(got).cpu().numpy()
    ref = sq.tf_queue_batches(n_rec, cap, batch, mad, nb, lambda n, k: sq.device_draws(seed, n, k))
    assert np.array_equal(got, ref)
    assert st.tolist() == [cap + nb * batch, nb]


@needs_gpu
def test_dequeue_and_gather_captured_in_a_graph():
    from air import _hip as H
    seed = 7
    q = torch.zeros(CAP, dtype=torch.int32, device="cuda")
    st = torch.zeros(2, dtype=torch.int64, device="cuda")
    picks = torch.zeros(BATCH, dtype=torch.int32, device="cuda")
    data = torch.arange(N_REC, device="cuda", dtype=torch.float32)[:, None].repeat(1, 8)
    out = torch.zeros(BATCH, 8, device="cuda")
    a = _struct(H, q, st, picks, seed)
    s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)   # noqa: E731
    H.check(H.lib().air_shuffle_batch_init(C.byref(a), s()))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    seen = []
    with torch.cuda.graph(g):
        for _ in range(3):                                             # three batches per replay
            H.check(H.lib().air_shuffle_batch_dequeue(C.byref(a), s()))
            torch.index_select(data, 0, picks, out=out)
            seen.append(out.clone())
    H.check(H.lib().air_shuffle_batch_init(C.byref(a), s()))          # capture consumed nothing; start over anyway
    rows = []
    for _ in range(5):
        g.replay()
        torch.cuda.synchronize()
        rows += [t[:, 0].cpu().numpy().astype(np.int64) for t in seen]
    ref = sq.tf_queue_batches(N_REC, CAP, BATCH, MIN_AFTER, 15, lambda n, k: sq.device_draws(seed, n, k))
    assert np.array_equal(np.stack(rows), ref)


@needs_gpu
@pytest.mark.parametrize("cap,batch,mad,n_rec,per_call", [(CAP, BATCH, MIN_AFTER, N_REC, 25), (100, 8, 50, 37, 7), (640, 64, 0, 1000, 1)])
def test_dequeue_many_equals_the_queue_model_pick_for_pick(cap, batch, mad, n_rec, per_call):
    from air import _hip as H
    calls, seed = 12, 0xabcdef12345
    q = torch.zeros(cap, dtype=torch.int32, device="cuda")
    st = torch.zeros(2, dtype=torch.int64, device="cuda")
    picks = torch.zeros(batch, dtype=torch.int32, device="cuda")
    many = torch.zeros(per_call, batch, dtype=torch.int32, device="cuda")
    a = _struct(H, q, st, picks, seed, cap, batch, mad, n_rec)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    H.check(H.lib().air_shuffle_batch_init(C.byref(a), s))
    got = []
    for k in range(calls):
        if k == 5:                                                     # the single-batch call continues the same sequence
            H.check(H.lib().air_shuffle_batch_dequeue(C.byref(a), s))
            got.append(picks.clone()[None])
        H.check(H.lib().air_shuffle_batch_dequeue_many(C.byref(a), per_call, many.data_ptr(), s))
        got.append(many.clone())
    torch.cuda.synchronize()
    got = torch.cat(got).cpu().numpy()
    nb = calls * per_call + 1
    ref = sq.tf_queue_batches(n_rec, cap, batch, mad, nb, lambda n, k: sq.device_draws(seed, n, k))
    assert np.array_equal(got, ref)
    assert st.tolist() == [cap + nb * batch, nb]


@needs_gpu
def test_batch_gather_copies_the_picked_records():
    from air import _hip as H
    g = torch.Generator(device="cuda").manual_seed(1)
    images = torch.rand(5000, 2500, device="cuda", generator=g)
    digits = torch.randint(0, 3, (5000,), device="cuda", dtype=torch.int32, generator=g)
    picks = torch.randint(0, 5000, (64,), device="cuda", dtype=torch.int32, generator=g)
    out_i, out_d = torch.zeros(64, 2500, device="cuda"), torch.zeros(64, dtype=torch.int32, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    H.check(H.lib().air_batch_gather(images.data_ptr(), digits.data_ptr(), picks.data_ptr(), out_i.data_ptr(), out_d.data_ptr(),
                                     64, 2500, s))
    assert torch.equal(out_i, images[picks.long()]) and torch.equal(out_d, digits[picks.long()])
    big = torch.rand(300, 128 * 128, device="cuda", generator=g)      # configs[3] canvases: several slices per row
    out_b = torch.zeros(64, 128 * 128, device="cuda")
    pk = torch.randint(0, 300, (64,), device="cuda", dtype=torch.int32, generator=g)
    H.check(H.lib().air_batch_gather(big.data_ptr(), None, pk.data_ptr(), out_b.data_ptr(), None, 64, 128 * 128, s))
    assert torch.equal(out_b, big[pk.long()])


@needs_gpu
def test_pipelined_queue_in_a_graph_delivers_the_same_batches():
    """multi_mnist.ShuffleBatchQueue.graph_hooks: the picks of a replay's batches made by its first launch, one row gather
    per step -- the batches that reach the consumer's buffers are, in order, the ones next_batch() delivers (= the queue
    model's)."""
    from multi_mnist import ShuffleBatchQueue
    n, D, B, steps, seed = 3000, 16, 8, 5, 21
    images = torch.arange(n, device="cuda", dtype=torch.float32)[:, None].repeat(1, D).contiguous()
    digits = (torch.arange(n, device="cuda", dtype=torch.int32) % 3).contiguous()

    def consumer():
        return torch.zeros(B, D, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda")
    oi, od = consumer()
    plain = ShuffleBatchQueue(images, digits, B, oi, od, seed=seed, min_after_dequeue=200)
    ref_rows = []
    for _ in range(4 * steps):
        plain.next_batch()
        ref_rows.append((oi[:, 0].clone(), od.clone()))
    pi, pd = consumer()
    piped = ShuffleBatchQueue(images, digits, B, pi, pd, seed=seed, min_after_dequeue=200)
    between, after = piped.graph_hooks(steps)
    assert after is None
    with pytest.raises(RuntimeError):
        piped.next_batch()
    with pytest.raises(ValueError):
        piped.graph_hooks(steps + 1)

    def capture():
        seen = []
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(steps):
                between(i)
                seen.append((pi[:, 0].clone(), pd.clone()))            # "the train step": reads the input buffers
        return g, seen
    g, seen = capture()
    got = []
    for _ in range(4):
        g.replay()
        got.append([(a.clone(), b.clone()) for a, b in seen])
    torch.cuda.synchronize()
    got = [ab for rep in got for ab in rep]
    for (ra, rb), (ga, gb) in zip(ref_rows, got):
        assert torch.equal(ra, ga) and torch.equal(rb, gb)
    model = sq.tf_queue_batches(n, 200 + 10 * B, B, 200, 4 * steps, lambda k, b: sq.device_draws(seed, k, b))
    assert np.array_equal(torch.stack([a for a, _ in got]).cpu().numpy().astype(np.int64), model)
    # a second capture (AIRModel captures again when a sampler-backward schedule switches) continues the sequence
    between, _ = piped.graph_hooks(steps)
    g2, seen2 = capture()
    g2.replay()
    torch.cuda.synchronize()
    more = sq.tf_queue_batches(n, 200 + 10 * B, B, 200, 5 * steps, lambda k, b: sq.device_draws(seed, k, b))[4 * steps:]
    assert np.array_equal(torch.stack([a for a, _ in seen2]).cpu().numpy().astype(np.int64), more)
