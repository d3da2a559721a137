"""The slice of polus/data.py that is on the data-parallel path: the rank-sharding rule
(polus/data.py:94-96, `Dataset.shard(num_shards=hvd.size(), index=hvd.local_rank())`):
element i goes to rank i mod size, BEFORE batching.  The generator -> tf.data machinery and
the pickle chunk cache are out of scope."""
from . import comm


def shard(iterable, num_shards=None, index=None):
    num_shards = comm.size() if num_shards is None else num_shards
    index = comm.local_rank() if index is None else index
    for i, item in enumerate(iterable):
        if i % num_shards == index:
            yield item


class ShardedBatches:
    """Re-iterable: shard the sample stream by rank, then batch (drop_remainder optional)."""

    def __init__(self, make_iter, batch_size, collate, drop_remainder=True, num_shards=None, index=None):
        self.make_iter, self.batch_size, self.collate = make_iter, batch_size, collate
        self.drop_remainder, self.num_shards, self.index = drop_remainder, num_shards, index

    def __iter__(self):
        buf = []
        for s in shard(self.make_iter(), self.num_shards, self.index):
            buf.append(s)
            if len(buf) == self.batch_size:
                yield self.collate(buf)
                buf = []
        if buf and not self.drop_remainder:
            yield self.collate(buf)
