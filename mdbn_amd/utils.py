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
                           exponent=1.0, repeats=10, shuffle=True, datadir='data', rng=None):
    """reference utils.py:77-119: load the table, preprocess, and return device-resident
    ``(train_set, validation_set)`` (SharedArray; validation None without hold-out)."""
    from .shared import shared
    n_data, n_cols, data = import_TCGA_data(datafile, datadir, dtype)
    train, validation = preprocess_table(data, holdout=holdout, clip=clip, transform_fn=transform_fn,
                                         exponent=exponent, repeats=repeats, shuffle=shuffle, rng=rng)
    return shared(train, borrow=True), (shared(validation, borrow=True) if validation is not None else None)


# Class post-processing of the joint layer's output (reference utils.py:121-176)

def remap_class(classified_samples, distance_matrix, n_classes):
    """Keep the ``n_classes`` most frequent classes and reassign every other class to a kept
    class near it in Hamming distance (reference utils.py:124-159, including its tie and
    overwrite order: the LAST admissible neighbour in ascending-distance order wins)."""
    def class_by_frequency(a):
        classes = range(int(numpy.max(a)) + 1)
        frequency = [numpy.sum(a == idx) for idx in classes]
        order = list(reversed(numpy.argsort(frequency).tolist()))
        return [{classes[i]: (r, frequency[i]) for r, i in enumerate(order)},
                {r: (classes[i], frequency[i]) for r, i in enumerate(order)}]

    def merge_classes(cmap, D, n_classes):
        new_map = {}
        n_initial_classes = D.shape[0]
        for i in range(n_classes):
            new_map[cmap[1][i][0]] = i
        for c in [cmap[1][i][0] for i in range(n_classes, n_initial_classes)]:
            for i in numpy.argsort(D[c]):
                r = cmap[0][i][0]
                if r < n_classes and r != c:
                    new_map[c] = r
        return new_map

    cmap = class_by_frequency(classified_samples)
    new_classification = merge_classes(cmap, distance_matrix, n_classes)
    return numpy.array([new_classification[i] for i in classified_samples])


def find_unique_classes(dbn_output):
    """Unique output-node patterns -> (class index per sample, Hamming distance matrix between
    the class patterns) (reference utils.py:162-176)."""
    from scipy.spatial import distance
    dbn_output = numpy.ascontiguousarray(dbn_output)
    class_representation = numpy.unique(dbn_output, axis=0)
    # the reference orders patterns by their raw bytes (numpy.void view); reproduce that order
    raw = class_representation.view(numpy.dtype((numpy.void, dbn_output.dtype.itemsize * dbn_output.shape[1])))
    class_representation = class_representation[numpy.argsort(raw.ravel())]
    distance_matrix = distance.cdist(class_representation, class_representation, metric='hamming')
    classified_samples = numpy.zeros((dbn_output.shape[0]))
    output_nodes = dbn_output.shape[1]
    for idx, pattern in enumerate(class_representation):
        classified_samples = classified_samples + \
            (numpy.sum(dbn_output == pattern, axis=1) == output_nodes) * idx
    return classified_samples, distance_matrix
