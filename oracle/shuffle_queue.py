"""Test infrastructure only: a numpy model of the reference's input queue,

    tf.train.shuffle_batch([image, digits], batch_size, capacity=10000 + 10 * batch_size, min_after_dequeue=10000,
                           num_threads)                                   (/root/reference/multi_mnist.py:240-249)

over string_input_producer([file], num_epochs) + ONE TFRecordReader (training.py:76-81): the record stream is the file
order, repeated every epoch.  TF 1.3's RandomShuffleQueue (random_shuffle_queue_op.cc, TryDequeueMany / DequeueLocked):
every dequeued element is  index = generator() % size;  take queue[index];  queue[index] = queue.back();  pop_back  -- a
uniform pick from what is resident; enqueue appends at the back and blocks at `capacity`; a dequeue waits until
size > min_after_dequeue.  The reader threads out-run a train step by orders of magnitude, so the queue is at capacity
whenever a batch is taken (the assumption of air_shuffle_batch_t in include/air_hip.h, stated there).

tf_queue_batches() is that queue written out literally (python list, pops and appends) and takes ANY uint32 source;
philox4x32_10() is the device's generator, so that model and kernel can be compared pick for pick."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
TAG = 0x53485546
MASK = 0xFFFFFFFF


def philox4x32_10(counter, key):
    c = [int(x) & MASK for x in counter]
    k0, k1 = int(key[0]) & MASK, int(key[1]) & MASK
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c


def device_draws(seed, n, batch):
    """the `batch` uint32 draws of dequeue number n (include/air_hip.h: air_shuffle_batch_t)"""
    out = []
    for g in range(batch // 4):
        out += philox4x32_10((n & MASK, (n >> 32) & MASK, g, TAG), (seed & MASK, (seed >> 32) & MASK))
    return out


def tf_queue_batches(n_records, capacity, batch, min_after_dequeue, num_batches, draws):
    """`num_batches` batches of record indices from the literal queue; draws(n, batch) -> the uint32s of batch n"""
    assert capacity - batch >= min_after_dequeue
    stream_pos = 0
    queue = []
    out = []
    for n in range(num_batches):
        while len(queue) < capacity:                      # the readers fill the queue to capacity before the consumer returns
            queue.append(stream_pos % n_records)
            stream_pos += 1
        r = draws(n, batch)
        picks = []
        for k in range(batch):
            assert len(queue) > min_after_dequeue
            idx = r[k] % len(queue)
            picks.append(queue[idx])
            queue[idx] = queue[-1]
            queue.pop()
        out.append(picks)
    return np.asarray(out, np.int64)


def parallel_dequeue(queue, batch, r, stream_pos, n_records):
    """ONE dequeue as the device kernel resolves it (csrc/air_input.hip: dequeue_batch) -- every pick on its own instead of
    the chain of swaps: idx_k = r_k mod (capacity - k) does not depend on the earlier picks, and the content of slot s at the
    time of pick t is found by walking the earlier picks backwards, V(s, t) = V(capacity - 1 - j, j) for the last j < t with
    idx_j == s (that pick moved the back of its time into s), else the original queue[s].  Pick k emits V(idx_k, k); the last
    pick on a surviving slot leaves V(back_k, k) there; the freed slots at the back are refilled from the stream.
    Returns (picks, new queue); tests hold it against the literal queue above."""
    cap = len(queue)
    idx = [int(r[k]) % (cap - k) for k in range(batch)]
    new = list(queue)
    picks = []
    for k in range(batch):
        s, sb = idx[k], cap - 1 - k
        for j in range(k - 1, -1, -1):
            if idx[j] == s:
                s = cap - 1 - j
            if idx[j] == sb:
                sb = cap - 1 - j
        picks.append(queue[s])
        if idx[k] < cap - batch and all(idx[j] != idx[k] for j in range(k + 1, batch)):
            new[idx[k]] = queue[sb]
    for t in range(batch):
        new[cap - batch + t] = (stream_pos + t) % n_records
    return picks, new
