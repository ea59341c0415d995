"""RandomStreams: seed holder standing in for Theano's MRG_RandomStreams
(reference rbm.py:40,92; dbn.py:39,114).  Each RBM that shares one RandomStreams draws
a distinct Philox stream id from it, so layers of a DBN use independent streams of the
same seed (the reference shares one MRG generator across layers, dbn.py:110-114,189,197)."""


class RandomStreams(object):
    def __init__(self, seed=12345):
        self.seed = int(seed)
        self._next_stream = 0

    def new_stream(self):
        s = self._next_stream
        self._next_stream += 1
        return s
