"""CPU: the host input pipeline (polus_amd/data.py) behaves as the reference's own data tests demand
(/root/reference/tests/test_data.py:28-345 -- order preservation, get_n_samples, cache `.index` / `.part` /
`.lookup` files on disk, pre_shuffle breaking the order at chunk level, merge counts 1000+500+721,
from_cached_index), plus the tf.data-like chain the tutorials build (map / cache / shuffle / batch / prefetch)
and the rank-sharding rule of polus/data.py:94-96."""
import os

import numpy as np
import pytest

from polus_amd.data import CachedDataLoader, CachedDataLoaderwLookup, DataLoader, Dataset

N = 1000


def gen_fn(n=N, text="dummy"):
    def source_gen():
        for i in range(n):
            yield {"id": i, "text": text}
    return source_gen


def in_order(it, key=lambda s: s["id"]):
    ok, n, last = True, 0, None
    for s in it:
        ok = ok and key(s) == n
        n += 1
        last = s
    return ok, n, last


def test_dataloader_preserves_order_and_counts():
    dl = DataLoader(gen_fn())
    ok, n, last = in_order(dl)
    assert ok and n == N and dl.get_n_samples() == N
    assert last["id"] == N - 1 and last["text"] == "dummy"
    assert dl.__name__ == "DataLoader_source_gen"
    ok, n, last = in_order(dl.to_tfDataset())          # re-iterable, same order through the Dataset
    assert ok and n == N
    with pytest.raises(ValueError):
        iter(DataLoader(lambda: [1, 2, 3]))             # a function that does not return a generator


def test_cached_dataloader_writes_index_and_parts(tmp_path):
    cache = str(tmp_path / "data")
    dl = CachedDataLoader(gen_fn(), cache_chunk_size=256, cache_folder=cache)
    ok, n, last = in_order(dl)
    assert ok and n == N and dl.get_n_samples() == N and last["text"] == "dummy"
    files = os.listdir(cache)
    assert f"{dl.cache_base_name}.index" in files
    assert len(dl.cache_index["files"]) == 4                       # 256 + 256 + 256 + 232
    for f in dl.cache_index["files"]:
        assert os.path.basename(f) in files
    assert dl.cache_index["n_samples"] == N and dl.cache_index["cache_chunk_size"] == 256
    # a second loader over the same generator finds the cache and does not run the generator again
    calls = {"n": 0}

    def source_gen():
        calls["n"] += 1
        yield from gen_fn()()
    CachedDataLoader(source_gen, cache_chunk_size=256, cache_folder=cache)
    assert calls["n"] == 0
    ok, n, _ = in_order(dl.to_tfDataset())
    assert ok and n == N


def test_cached_dataloader_cleans_up_after_a_failing_generator(tmp_path):
    cache = str(tmp_path / "data")

    def bad_gen():
        for i in range(100):
            if i == 70:
                raise RuntimeError("boom")
            yield {"id": i}
    with pytest.raises(RuntimeError):
        CachedDataLoader(bad_gen, cache_chunk_size=16, cache_folder=cache)
    assert os.listdir(cache) == []


def test_cached_dataloader_custom_generator_order(tmp_path):
    def new_gen():
        for s in gen_fn()():
            s["new_entry"] = s["id"] * 2
            yield s
    dl = CachedDataLoader(new_gen, cache_chunk_size=16, cache_folder=str(tmp_path))
    ok, n, last = in_order(dl, key=lambda s: s["id"] if s["new_entry"] == 2 * s["id"] else -1)
    assert ok and n == N and last["new_entry"] == 2 * last["id"]


def test_pre_shuffle_breaks_order_at_chunk_level(tmp_path):
    dl = CachedDataLoader(gen_fn(), cache_chunk_size=16, cache_folder=str(tmp_path)).pre_shuffle()
    seen = [s["id"] for s in dl]
    assert seen != list(range(N)) and sorted(seen) == list(range(N)) and dl.get_n_samples() == N
    # samples inside a chunk keep their order: every run of 16 starts at a multiple of 16 and is consecutive
    for k in range(0, N - 16, 16):
        run = seen[k:k + 16]
        if run[0] % 16 == 0 and run[0] + 16 <= N:
            assert run == list(range(run[0], run[0] + 16))
    assert [s["id"] for s in dl] != seen                           # a fresh order on every pass


def test_merge_and_from_cached_index(tmp_path):
    cache = str(tmp_path)

    def source_gen_1():
        yield from gen_fn(1000, "dummy")()

    def source_gen_2():
        yield from gen_fn(500, "dummy2")()

    def source_gen_3():
        yield from gen_fn(721, "dummy3")()
    dls = [CachedDataLoader(g, cache_chunk_size=64, cache_folder=cache) for g in (source_gen_1, source_gen_2, source_gen_3)]
    merged = CachedDataLoader.merge(*dls)
    assert merged.get_n_samples() == 1000 + 500 + 721
    ordered = [s["text"] for s in merged]
    assert ordered == ["dummy"] * 1000 + ["dummy2"] * 500 + ["dummy3"] * 721
    merged.pre_shuffle()
    ok, n, _ = in_order(merged)
    assert not ok and n == 2221
    path = dls[0].cache_index_path
    again = CachedDataLoader.from_cached_index(path)
    ok, n, last = in_order(again)
    assert ok and n == 1000 and last["text"] == "dummy" and again.cache_index_path == path
    copy = again.deep_copy(suffix="copy")
    assert os.path.exists(copy.cache_index_path) and copy.cache_index_path.endswith("_copy.index")


def test_lookup_loader_and_conversion(tmp_path):
    cache = str(tmp_path)
    data = {"a": 1, "b": 2, "c": 3}
    dl = CachedDataLoaderwLookup(gen_fn(), lookup_data=data, cache_chunk_size=256, cache_folder=cache)
    assert sum(1 for _ in dl) == N and dl.get_lookup_data()["c"] == 3
    assert f"{dl.cache_base_name}.lookup" in os.listdir(cache) and f"{dl.cache_base_name}.index" in os.listdir(cache)
    with pytest.raises(ValueError):
        CachedDataLoaderwLookup(gen_fn(), cache_folder=cache)

    def other_gen():
        yield from gen_fn(300)()
    conv = CachedDataLoader(other_gen, cache_chunk_size=128, cache_folder=cache).add_lookup_data(["x", "y"])
    assert isinstance(conv, CachedDataLoaderwLookup) and conv.get_lookup_data() == ["x", "y"] and conv.get_n_samples() == 300
    base = os.path.splitext(os.path.basename(conv.cache_index_path))[0]
    assert f"{base}.lookup" in os.listdir(cache)
    conv.clean()
    assert f"{base}.lookup" not in os.listdir(cache) and f"{base}.index" not in os.listdir(cache)


def test_dataset_chain_of_the_tutorial():
    """tutorials/classifier_example.py:29-42: map -> cache -> shuffle -> batch(drop_remainder) -> prefetch."""
    x = np.arange(1000 * 4, dtype=np.uint8).reshape(1000, 2, 2)
    y = np.arange(1000) % 10

    def generator():
        for i in range(len(x)):
            yield {"x": x[i], "y": y[i]}
    calls = {"n": 0}

    def normalize(d):
        calls["n"] += 1
        return d["x"].astype(np.float32) / 255.0, np.int32(d["y"])
    ds = DataLoader(generator).to_tfDataset().map(normalize).cache().shuffle(1000, seed=3).batch(128, drop_remainder=True)
    ds = ds.prefetch(-1, to_device=False)
    b1 = list(ds)
    assert len(b1) == 7 and b1[0][0].shape == (128, 2, 2) and b1[0][0].dtype == np.float32 and b1[0][1].dtype == np.int32
    ids1 = np.concatenate([b[1] for b in b1])
    b2 = list(ds)
    assert calls["n"] == 1000                                       # cache(): the map ran once
    assert not np.array_equal(ids1, np.concatenate([b[1] for b in b2]))   # reshuffled each epoch
    test = DataLoader(generator).to_tfDataset().map(normalize).batch(128)
    assert [b[0].shape[0] for b in test] == [128] * 7 + [104]
    with pytest.raises(TypeError):
        len(test)                                                    # unknown cardinality until counted


def test_dataset_shard_rule_and_lengths():
    ds = Dataset(lambda: iter(range(10)), 10)
    assert list(ds.shard(4, 1)) == [1, 5, 9] and len(ds.shard(4, 1)) == 3       # polus/data.py:94-96
    assert len(ds.batch(4)) == 3 and len(ds.batch(4, drop_remainder=True)) == 2
    assert list(ds.take(3)) == [0, 1, 2] and list(ds.repeat(2)) == list(range(10)) * 2
    got = list(ds.prefetch(3, to_device=False))
    assert got == list(range(10))

    def failing():
        yield 1
        raise KeyError("producer error")
    with pytest.raises(KeyError):
        list(Dataset(failing).prefetch(2, to_device=False))
