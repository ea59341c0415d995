"""Host helpers on the hot loop (reference src/utils.py)."""
import numpy


def get_minibatches_idx(n, batch_size, shuffle=False, rng=None):
    """Minibatch index lists, as reference utils.py:54-75: int32 arange, optional shuffle,
    ``n // batch_size`` full batches plus a ragged tail.  The reference shuffles with the
    unseeded global ``numpy.random`` (utils.py:62); pass ``rng`` (a RandomState) for
    reproducible runs."""
    idx_list = numpy.arange(n, dtype="int32")

    if shuffle:
        (rng if rng is not None else numpy.random).shuffle(idx_list)

    minibatches = []
    minibatch_start = 0
    for i in range(n // batch_size):
        minibatches.append(idx_list[minibatch_start:minibatch_start + batch_size])
        minibatch_start += batch_size

    if minibatch_start != n:
        minibatches.append(idx_list[minibatch_start:])

    return range(len(minibatches)), minibatches
