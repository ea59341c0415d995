"""Host helpers around the hot loop, with the signatures of the reference's src/utils.py:
``get_minibatches_idx`` (on the loop), the TSV loader / preprocessing that feeds it
(``import_TCGA_data``, ``load_n_preprocess_data``) and the class post-processing that
consumes the joint layer's output (``find_unique_classes``, ``remap_class``).  Pure numpy /
scipy on the host, as in the reference; nothing here touches the device except the final
``shared(...)`` upload of the train / validation matrices."""
import gzip
import os

import numpy


def get_minibatches_idx(n, batch_size, shuffle=False, rng=None):
    """Minibatch index lists with the contract of reference utils.py:54-75: the int32 row numbers
    0..n-1, optionally shuffled, cut into ``n // batch_size`` full batches plus a ragged tail; returns
    ``(range(number of batches), list of index arrays)``.  The reference shuffles with the unseeded
    global ``numpy.random`` (utils.py:62); pass ``rng`` (a RandomState) for reproducible runs."""
    order = numpy.arange(n, dtype="int32")
    if shuffle:
        (numpy.random if rng is None else rng).shuffle(order)
    pieces = numpy.split(order, numpy.arange(batch_size, n, batch_size))
    batches = [p for p in pieces if len(p)]
    return range(len(batches)), batches


def import_TCGA_data(file, datadir, dtype):
    """Tab-separated table with a header row and a label column (reference utils.py:34-52):
    one COLUMN per person.  Returns ``(n_persons, n_persons, data)`` exactly as the reference
    does (its first two results are both the column count)."""
    path = os.path.join(datadir, file)
    opener = gzip.open if file.endswith('.gz') else open
    with opener(path, 'rt') as f:
        ncols = len(f.readline().split('\t'))
    data = numpy.loadtxt(path, dtype=dtype, delimiter='\t', skiprows=1, usecols=range(1, ncols),
                         ndmin=2)
    return (data.shape[1], ncols - 1, data)


def preprocess_table(data, holdout=0.1, clip=None, transform_fn=None, exponent=1.0, repeats=10,
                     shuffle=True, rng=None):
    """The array half of reference utils.py:77-119: optional transform, z-score each
    measurement over the population, drop measurements with a NaN, transpose to one ROW per
    person, clip, replicate rows, split off the hold-out set.  Returns ``(train, validation)``
    host arrays (validation is None without hold-out).

    Kept quirk: after ``numpy.repeat`` the split still draws indices from
    ``range(n_persons)``, so with ``repeats > 1`` only the first ``n_persons`` rows of the
    replicated matrix are used (reference utils.py:103-110)."""
    from scipy import stats
    n_cols = data.shape[1]
    if transform_fn is not None:
        data = transform_fn(data, exponent)
    zdata = stats.zscore(data, axis=1)
    zdata = zdata[~numpy.isnan(zdata).any(axis=1)].T
    if clip is not None:
        zdata = numpy.clip(zdata, clip[0], clip[1])
    if repeats > 1:
        zdata = numpy.repeat(zdata, repeats=repeats, axis=0)
    validation_set_size = int(n_cols * holdout)
    _, indexes = get_minibatches_idx(n_cols, n_cols - validation_set_size, shuffle=shuffle, rng=rng)
    train = zdata[indexes[0]]
    validation = zdata[indexes[1]] if validation_set_size > 0 else None
    return train, validation


def load_n_preprocess_data(datafile, dtype='float32', holdout=0.1, clip=None, transform_fn=None,
                           exponent=1.0, repeats=10, shuffle=True, datadir='data', rng=None, resident="device"):
    """reference utils.py:77-119: load the table, preprocess, and return ``(train_set, validation_set)``
    (SharedArray; validation None without hold-out).  ``resident``: where the TRAINING table lives -- "device"
    (as the reference's theano.shared, utils.py:113-115), "host" (pinned memory, minibatch rows gathered over PCIe
    one step ahead: shared.HostTable) or "auto"; the validation set is always device-resident."""
    from .shared import shared
    n_data, n_cols, data = import_TCGA_data(datafile, datadir, dtype)
    train, validation = preprocess_table(data, holdout=holdout, clip=clip, transform_fn=transform_fn,
                                         exponent=exponent, repeats=repeats, shuffle=shuffle, rng=rng)
    return shared(train, borrow=True, resident=resident), (shared(validation, borrow=True) if validation is not None else None)


# Class post-processing of the joint layer's output (reference utils.py:121-176)

def remap_class(classified_samples, distance_matrix, n_classes):
    """Keep the ``n_classes`` most frequent classes (relabelled 0, 1, ... by falling frequency) and fold
    every other class into a kept one (specification: reference utils.py:124-159).

    The reference walks a dropped class's neighbours by rising Hamming distance and overwrites its
    choice at every admissible one, so the LAST admissible neighbour wins (the farthest, not the
    nearest); "admissible" means the neighbour's frequency rank is below ``n_classes`` and differs
    from the dropped class's own ID (a rank compared with an ID -- kept as is).  Ties in frequency
    rank the higher class ID first (a reversed ascending argsort)."""
    labels = numpy.asarray(classified_samples)
    counts = numpy.array([numpy.sum(labels == c) for c in range(int(numpy.max(labels)) + 1)])
    by_rank = numpy.argsort(counts)[::-1]                    # class IDs, most frequent first
    rank_of = numpy.empty(len(by_rank), dtype=int)
    rank_of[by_rank] = numpy.arange(len(by_rank))
    n_initial = distance_matrix.shape[0]
    table = numpy.full(max(n_initial, len(by_rank)), -1, dtype=int)
    table[by_rank[:n_classes]] = numpy.arange(min(n_classes, len(by_rank)))
    for c in by_rank[n_classes:n_initial]:
        ranks = rank_of[numpy.argsort(distance_matrix[c])]
        admissible = ranks[(ranks < n_classes) & (ranks != c)]
        if len(admissible):
            table[c] = admissible[-1]
    merged = table[labels.astype(int)]
    if (merged < 0).any():
        raise KeyError("class %d has no admissible neighbour" % int(labels[merged < 0][0]))
    return merged


def find_unique_classes(dbn_output):
    """Distinct output-node patterns -> (class index of every sample, Hamming distance matrix between
    the class patterns) (reference utils.py:162-176).  Classes are numbered in the order of the
    patterns' raw bytes, which is what the reference's ``numpy.unique`` over a void view yields."""
    from scipy.spatial.distance import cdist
    rows = numpy.ascontiguousarray(dbn_output)
    as_bytes = rows.view(numpy.dtype((numpy.void, rows.dtype.itemsize * rows.shape[1]))).ravel()
    order = numpy.argsort(as_bytes, kind='stable')
    ranked = rows[order]
    opens_class = numpy.ones(len(rows), dtype=bool)
    opens_class[1:] = (ranked[1:] != ranked[:-1]).any(axis=1)
    labels = numpy.empty(len(rows), dtype=numpy.float64)
    labels[order] = numpy.cumsum(opens_class) - 1
    patterns = ranked[opens_class]
    return labels, cdist(patterns, patterns, metric='hamming')
