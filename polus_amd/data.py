"""polus/data.py drop-in: the input side of the training step.

* `DataLoader` / `CachedDataLoader` / `CachedDataLoaderwLookup` (polus/data.py:49-494): python sample
  generators, optionally materialised once into pickle chunk files (`<base>_NNNN.part`) listed by a JSON
  `<base>.index` -- the reference's on-disk format, so caches written by either side are readable by the
  other -- with chunk-level `pre_shuffle`, `merge`, `from_cached_index`, lookup side-car.
* `to_tfDataset()` returns a `Dataset`: the subset of the tf.data chain the reference's scripts use
  (map / cache / shuffle / batch / prefetch / shard / take / repeat), lazily evaluated on the host.
  The rank-sharding rule is the reference's (polus/data.py:94-96): element i goes to rank
  i mod size, BEFORE batching.
* `Dataset.prefetch(n)` on a batched dataset is where the MI355X side differs: a background thread
  collates the next batches into PINNED host buffers and queues their host-to-device copies on a
  side HIP stream, n batches ahead, so `next(iterator)` in the trainer's loop (polus/training.py:310) hands
  over tensors that are already in HBM and the 16 ms step never waits for the PCIe transfer.

Pickle is only ever applied to files this module wrote (the cache directory is the user's own)."""
import json
import os
import pickle
import queue
import random
import threading
import types

import numpy as np

from . import comm
from .context import PolusContext, logger

AUTOTUNE = -1


# ------------------------------------------------------------------------------- tf.data-like chain
def _collate(samples):
    """list of samples -> one batch, same structure (dict / tuple / leaf), leaves stacked on axis 0."""
    first = samples[0]
    if isinstance(first, dict):
        return {k: _collate([s[k] for s in samples]) for k in first}
    if isinstance(first, (tuple, list)):
        return type(first)(_collate([s[i] for s in samples]) for i in range(len(first)))
    if isinstance(first, (str, bytes)):
        return np.asarray(samples, dtype=object)
    try:
        import torch
        if isinstance(first, torch.Tensor):
            return torch.stack(samples, 0)
    except ImportError:
        pass
    return np.stack([np.asarray(s) for s in samples], 0)


def _tree_map(f, x):
    if isinstance(x, dict):
        return {k: _tree_map(f, v) for k, v in x.items()}
    if isinstance(x, (tuple, list)):
        return type(x)(_tree_map(f, v) for v in x)
    return f(x)


class Dataset:
    """A lazily evaluated chain over a re-iterable source.  Every transformation returns a new Dataset;
    iterating runs the chain from the source."""

    def __init__(self, make_iter, length=None):
        self._make_iter = make_iter
        self._length = length          # None = unknown cardinality (len() raises TypeError, like tf's -2)

    @classmethod
    def from_generator(cls, gen_fn, length=None):
        return cls(lambda: iter(gen_fn()), length)

    def __iter__(self):
        return self._make_iter()

    def __len__(self):
        if self._length is None:
            raise TypeError("unknown cardinality")
        return self._length

    def cardinality(self):
        return -2 if self._length is None else self._length

    # ---- transformations
    def map(self, f, num_parallel_calls=None, **_):
        return Dataset(lambda: (f(x) for x in self), self._length)

    def shard(self, num_shards, index):
        """tf.data.Dataset.shard: every num_shards-th element starting at `index`."""
        n = None if self._length is None else max(0, (self._length - index + num_shards - 1) // num_shards)
        return Dataset(lambda: (x for i, x in enumerate(self) if i % num_shards == index), n)

    def cache(self, filename=""):
        store = {}

        def it():
            if "data" in store:
                return iter(store["data"])

            def fill():
                buf = []
                for x in self:
                    buf.append(x)
                    yield x
                store["data"] = buf
            return fill()
        return Dataset(it, self._length)

    def shuffle(self, buffer_size, seed=None, reshuffle_each_iteration=True):
        """Reservoir-style shuffle buffer as tf.data's: fill `buffer_size`, then emit a random slot and refill."""
        state = {"epoch": 0}

        def it():
            rng = random.Random(None if seed is None else seed + (state["epoch"] if reshuffle_each_iteration else 0))
            state["epoch"] += 1
            buf = []
            for x in self:
                if len(buf) < buffer_size:
                    buf.append(x)
                    continue
                j = rng.randrange(len(buf))
                buf[j], x = x, buf[j]
                yield x
            rng.shuffle(buf)
            yield from buf
        return Dataset(it, self._length)

    def batch(self, batch_size, drop_remainder=False, **_):
        def it():
            buf = []
            for x in self:
                buf.append(x)
                if len(buf) == batch_size:
                    yield _collate(buf)
                    buf = []
            if buf and not drop_remainder:
                yield _collate(buf)
        n = None
        if self._length is not None:
            n = self._length // batch_size if drop_remainder else (self._length + batch_size - 1) // batch_size
        return Dataset(it, n)

    def take(self, count):
        def it():
            for i, x in enumerate(self):
                if i >= count:
                    return
                yield x
        return Dataset(it, None if self._length is None else min(self._length, count))

    def repeat(self, count=None):
        def it():
            k = 0
            while count is None or k < count:
                yield from self
                k += 1
        return Dataset(it, None if (count is None or self._length is None) else self._length * count)

    def prefetch(self, buffer_size=AUTOTUNE, to_device=None):
        """Produce elements `buffer_size` ahead on a background thread.  With a GPU visible (or
        to_device=True) numeric leaves are staged through pinned host memory and copied to HBM on a side
        stream, so the consumer receives device tensors whose copies were queued while the previous
        steps ran; to_device=False keeps host arrays."""
        depth = 2 if buffer_size in (AUTOTUNE, None) or buffer_size < 1 else int(buffer_size)
        return Dataset(lambda: _Prefetcher(self, depth, to_device), self._length)


class _Prefetcher:
    """Iterator: a daemon thread walks the upstream iterator, stages each element and hands it over
    through a bounded queue.  Device staging: leaf -> pinned host tensor -> cuda(non_blocking) on a
    private stream, with an event the consumer's stream waits on before first use."""
    _END = object()

    def __init__(self, upstream, depth, to_device):
        self._q = queue.Queue(maxsize=depth)
        self._device = None
        self._stream = None
        if to_device is None or to_device:
            try:
                import torch
                if torch.cuda.is_available():
                    self._device = torch.device("cuda", torch.cuda.current_device())
                    self._stream = torch.cuda.Stream(device=self._device)
                elif to_device:
                    raise RuntimeError("prefetch(to_device=True) needs a GPU")
            except ImportError:
                if to_device:
                    raise
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, args=(upstream,), daemon=True)
        self._thread.start()

    def _stage(self, x):
        if self._device is None:
            return x, None
        import torch

        def leaf(a):
            if isinstance(a, torch.Tensor):
                t = a
            elif isinstance(a, np.ndarray) and a.dtype != object and a.dtype.kind in "biuf":
                t = torch.from_numpy(np.ascontiguousarray(a))
            else:
                return a
            if t.dtype == torch.float64:
                t = t.float()
            elif t.dtype == torch.int64:
                t = t.to(torch.int32) if t.numel() == 0 or int(t.abs().max()) < 2 ** 31 else t
            if t.device.type == "cuda":
                return t
            return t.pin_memory().to(self._device, non_blocking=True)
        with torch.cuda.stream(self._stream):
            y = _tree_map(leaf, x)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return y, ev

    def _run(self, upstream):
        try:
            if self._device is not None:
                import torch
                torch.cuda.set_device(self._device)
            for x in upstream:
                if not self._put(self._stage(x)):
                    return
            self._put((self._END, None))
        except BaseException as e:      # surface producer errors in the consumer
            self._put((e, None))

    def _put(self, item):
        """Blocking put that gives up once the consumer is gone (an abandoned iteration -- early_stop, take() --
        must not leave this thread parked on a full queue holding pinned and device batches)."""
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def __iter__(self):
        return self

    def __next__(self):
        x, ev = self._q.get()
        if x is self._END:
            self._q.put((self._END, None))
            raise StopIteration
        if isinstance(x, BaseException):
            raise x
        if ev is not None:
            import torch
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            # The batch was allocated under the prefetch stream: tell the caching allocator that the consumer's
            # stream uses it too, or the block could be handed to the next prefetch copy (and overwritten) while
            # kernels already queued on the compute stream -- embed_bwd reads the ids late in backward -- have
            # not read it yet.

            def mark(a):
                if isinstance(a, torch.Tensor) and a.is_cuda:
                    a.record_stream(cur)
                return a
            _tree_map(mark, x)
        return x

    def close(self):
        """Stops the producer thread and drops what it has staged."""
        self._stop.set()
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass

    def __del__(self):
        self._stop.set()


# ------------------------------------------------------------------------------- rank sharding
def shard(iterable, num_shards=None, index=None):
    """polus/data.py:94-96: element i goes to rank i mod size, before batching."""
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


# ------------------------------------------------------------------------------- loaders
class DataLoader:
    """polus/data.py:49-136: wraps a generator FUNCTION (or a generator) of sample dictionaries."""

    def __init__(self, sample_generator, magic_k=10):
        super().__init__()
        self.name = sample_generator.__name__ if sample_generator is not None else "None"
        self.sample_generator = sample_generator
        self.magic_k = magic_k

    def to_tfDataset(self):
        """The sample stream as a `Dataset`; sharded by rank when data parallelism is on (:94-96)."""
        n = getattr(self, "n_samples", None)
        ds = Dataset(lambda: iter(self), n)
        if PolusContext().is_horovod_enabled():
            ds = ds.shard(num_shards=comm.size(), index=comm.local_rank())
        return ds

    def set_name(self, _name):
        self.name = _name

    @property
    def __name__(self):
        return f"{self.__class__.__name__}_{self.name}"

    def __iter__(self):
        if isinstance(self.sample_generator, types.GeneratorType):
            return iter(self.sample_generator)
        it = self.sample_generator()
        if not isinstance(it, types.GeneratorType):
            raise ValueError("The sample_generator that was set in the DataLoader was a function that did not "
                             "return an generator, it must return a generator")
        return it

    def get_n_samples(self):
        """Counts once (one pass over the generator) and remembers."""
        if not hasattr(self, "n_samples"):
            logger.info("this dataset does not have the number of samples in cache so it will take some time to counting")
            self.n_samples = sum(1 for _ in self)
        return self.n_samples


class _ChunkStore:
    """The on-disk side of a cached loader: `<base>.index` (JSON: files, cache_chunk_size, n_samples
    [, lookup_file]) + `<base>_NNNN.part` pickle files of up to cache_chunk_size samples each."""

    @staticmethod
    def read_index(path):
        with open(path, "r") as f:
            return json.load(f)

    @staticmethod
    def write_chunks(gen_fn, base_path, chunk_size):
        files, n, buf = [], 0, []

        def flush():
            path = f"{base_path}_{len(files):04}.part"
            with open(path, "wb") as f:
                pickle.dump(buf, f)
            files.append(path)
        for sample in gen_fn():
            n += 1
            buf.append(sample)
            if len(buf) >= chunk_size:
                flush()
                buf = []
        if buf:
            flush()
        return {"files": files, "cache_chunk_size": chunk_size, "n_samples": n}

    @staticmethod
    def iterate(index, shuffle_blocks):
        order = list(range(len(index["files"])))
        if shuffle_blocks:
            random.shuffle(order)
        for k in order:
            with open(index["files"][k], "rb") as f:
                yield from pickle.load(f)


class CachedDataLoader(DataLoader):
    """polus/data.py:138-410.  First construction runs the generator once and writes the chunks; later
    constructions with the same generator name / chunk size / identifier find the `.index` and read
    the chunks instead.  If caching fails half way, the files written so far are removed."""

    def __init__(self, sample_generator=None, clean_up_function=None, cache_additional_identifier="",
                 cache_chunk_size=8192, cache_folder=os.path.join(".polus_cache", "data"), cache_index=None, **kwargs):
        assert sample_generator is not None or cache_index is not None
        self.cache_folder = cache_folder
        self.cache_chunk_size = cache_chunk_size
        self.cache_additional_identifier = cache_additional_identifier
        self.shuffle_blocks = False
        self.clean_up_function = clean_up_function
        self.cache_index = cache_index
        self.cache_index_path = cache_index.get("cache_index_path") if cache_index is not None else None
        try:
            gen = self._build_sample_generator(sample_generator)
            super().__init__(sample_generator=gen, **kwargs)
        except Exception:
            if cache_index is None:
                logger.info("An error has occured so all the created files will be deleted")
                self.clean()
            raise

    @property
    def __name__(self):
        return f"{self.__class__.__name__}_{self.name}"

    @staticmethod
    def read_index(file_path):
        return _ChunkStore.read_index(file_path)

    @classmethod
    def from_cached_index(cls, index_path):
        info = cls.read_index(index_path)
        info["cache_index_path"] = index_path
        return cls(cache_index=info)

    @classmethod
    def _merged_index(cls, loaders):
        assert len(loaders) > 1
        info = {"files": [], "cache_chunk_size": 0, "n_samples": 0}
        for dl in loaders:
            idx = cls.read_index(dl.cache_index_path)
            info["n_samples"] += idx["n_samples"]
            info["files"].extend(idx["files"])
            info["cache_chunk_size"] = max(info["cache_chunk_size"], idx["cache_chunk_size"])
        return info

    @classmethod
    def merge(cls, *cache_dataloaders):
        return cls(cache_index=cls._merged_index(cache_dataloaders))

    def _cache_base_name(self, sample_generator):
        prefix = f"{self.cache_additional_identifier}_" if self.cache_additional_identifier != "" else ""
        return f"{prefix}_chunk{self.cache_chunk_size}_{sample_generator.__name__}"

    def write_index_file(self, index_info):
        with open(self.cache_index_path, "w") as f:
            json.dump(index_info, f)

    def _build_sample_generator(self, sample_generator):
        if self.cache_index is None:
            os.makedirs(self.cache_folder, exist_ok=True)
            self.cache_base_name = self._cache_base_name(sample_generator)
            self.cache_base_path = os.path.join(self.cache_folder, self.cache_base_name)
            self.cache_index_path = f"{self.cache_base_path}.index"
            if not os.path.exists(self.cache_index_path):
                logger.info(f"DataLoader will store the samples in {self.cache_base_path}, with a max_sample per file "
                            f"of {self.cache_chunk_size}, this may take a while")
                self.cache_index = {"files": []}       # what clean() removes if the generator raises half way
                info = _ChunkStore.write_chunks(lambda: self._tracked(sample_generator), self.cache_base_path, self.cache_chunk_size)
                self.write_index_file(info)
            else:
                logger.info("We found a compatible cache file for this DataLoader")
            if self.clean_up_function is not None:
                logger.info("Executing the clean up function after the cached dataset was created")
                self.clean_up_function()
            self.cache_index = self.read_index(self.cache_index_path)
        return self._generator_from_index()

    def _tracked(self, sample_generator):
        """The user's generator, with the part files written so far registered for clean()."""
        n = 0
        for s in sample_generator():
            n += 1
            if (n - 1) % self.cache_chunk_size == 0:
                self.cache_index["files"].append(f"{self.cache_base_path}_{len(self.cache_index['files']):04}.part")
            yield s

    def _generator_from_index(self):
        self.n_samples = self.cache_index["n_samples"]
        self.cache_chunk_size = self.cache_index["cache_chunk_size"]
        logger.info(f"Total number of samples in dataset: {self.n_samples}")

        def generator():
            yield from _ChunkStore.iterate(self.cache_index, self.shuffle_blocks)
        return generator

    def clean(self):
        idx = getattr(self, "cache_index", None)
        if idx is not None and "files" in idx:
            for file in idx["files"]:
                if os.path.exists(file):
                    os.remove(file)
        p = getattr(self, "cache_index_path", None)
        if p is not None and os.path.exists(p):
            os.remove(p)

    def pre_shuffle(self):
        """Chunk files are read in a fresh random order on every pass (samples inside a chunk keep their order)."""
        self.shuffle_blocks = True
        return self

    def add_lookup_data(self, lookup_object):
        lookup_file = f"{os.path.splitext(self.cache_index_path)[0]}.lookup"
        with open(lookup_file, "wb") as f:
            pickle.dump(lookup_object, f)
        return self.add_lookup_data_path(lookup_file)

    def add_lookup_data_path(self, lookup_data_path):
        assert os.path.exists(lookup_data_path)
        self.cache_index["lookup_file"] = lookup_data_path
        with open(self.cache_index_path, "w") as f:
            json.dump({k: v for k, v in self.cache_index.items() if k != "cache_index_path"}, f)
        return CachedDataLoaderwLookup.from_cached_index(self.cache_index_path)

    def deep_copy(self, path=None, suffix=None):
        if path is None:
            suffix = "copy" if suffix is None else suffix
            path = f"{os.path.splitext(self.cache_index_path)[0]}_{suffix}.index"
        with open(path, "w") as f:
            json.dump({k: v for k, v in self.cache_index.items() if k != "cache_index_path"}, f)
        self.cache_index_path = path
        return self


class CachedDataLoaderwLookup(CachedDataLoader):
    """polus/data.py:413-494: a cached loader with one pickled side-car object (`<base>.lookup`)."""

    def __init__(self, *args, lookup_data=None, cache_index=None, **kwargs):
        if cache_index is not None and "lookup_file" in cache_index:
            lookup_data = self._load_lookup_data(cache_index["lookup_file"])
        if lookup_data is None:
            raise ValueError("Do not use CachedDataLoaderwLookup without setting a lookup_data, instead use CachedDataLoader")
        self.lookup_data = lookup_data
        super().__init__(*args, cache_index=cache_index, **kwargs)

    def get_lookup_data(self):
        return self.lookup_data

    @staticmethod
    def _load_lookup_data(lookup_file):
        with open(lookup_file, "rb") as f:
            return pickle.load(f)

    @classmethod
    def merge(cls, *cache_dataloaders):
        info = cls._merged_index(cache_dataloaders)
        lookup = []
        for dl in cache_dataloaders:
            lookup.extend(cls._load_lookup_data(cls.read_index(dl.cache_index_path)["lookup_file"]))
        return cls(cache_index=info, lookup_data=lookup)

    def clean(self):
        idx = getattr(self, "cache_index", None)
        if idx is not None and "lookup_file" in idx and os.path.exists(idx["lookup_file"]):
            os.remove(idx["lookup_file"])
        super().clean()

    def write_index_file(self, index_info):
        lookup_file = f"{self.cache_base_path}.lookup"
        with open(lookup_file, "wb") as f:
            pickle.dump(self.lookup_data, f)
        index_info["lookup_file"] = lookup_file
        with open(self.cache_index_path, "w") as f:
            json.dump(index_info, f)
