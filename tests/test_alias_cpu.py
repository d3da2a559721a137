"""CPU: the `polus` import path.  The import block of the reference's tutorial
(/root/reference/tutorials/classifier_example.py:1-5, minus `import tensorflow`) and of its other user-facing
modules resolves, unchanged, to the MI355X engine's classes."""


def test_tutorial_import_block_runs_unchanged():
    from polus.training import ClassifierTrainer
    from polus.metrics import MacroF1Score
    from polus.callbacks import LossSmoothCallback, ValidationDataCallback, ConsoleLogCallback, TimerCallback, EarlyStop
    from polus.data import DataLoader
    from polus.models import SequentialPolusClassifier
    import polus_amd.training, polus_amd.models, polus_amd.data
    assert ClassifierTrainer is polus_amd.training.ClassifierTrainer
    assert SequentialPolusClassifier is polus_amd.models.SequentialPolusClassifier
    assert DataLoader is polus_amd.data.DataLoader
    assert all(callable(c) for c in (MacroF1Score, LossSmoothCallback, ValidationDataCallback, ConsoleLogCallback, TimerCallback, EarlyStop))


def test_every_reference_module_on_the_path_has_an_alias():
    import importlib
    for name, symbols in {
        "polus": ["PolusContext", "logger"],
        "polus.training": ["BaseTrainer", "ClassifierTrainer"],
        "polus.ir.training": ["EfficientDenseRetrievalTrainer"],
        "polus.models": ["PolusModel", "SavableModel", "PolusClassifier", "SequentialPolusClassifier", "TFBertSplited",
                         "split_bert_model", "split_bert_model_from_checkpoint", "load_model", "from_config"],
        "polus.layers": ["CRF"],
        "polus.losses": ["weighted_softmax_cross_entropy_from_logits", "weighted_sigmoid_cross_entropy_from_logits"],
        "polus.schedulers": ["warmup_scheduler"],
        "polus.metrics": ["IMetric", "MacroF1Score"],
        "polus.callbacks": ["CallbackCoordinator", "SaveModelCallback", "Profiler"],
        "polus.data": ["DataLoader", "CachedDataLoader", "CachedDataLoaderwLookup"],
        "polus.core": ["get_jit_compile", "set_jit_compile", "find_dtype_and_shapes", "execute_if"],
        "polus.utils": ["flatten_dict", "merge_dicts", "Singleton", "complex_json_serializer", "complex_json_deserializer"],
        "polus.hpo": ["HPOContext", "parameter"],
        "polus.ner.models": ["baselineNER_MLP_CRF", "baselineNER_MLP_Dropout_CRF"],
        "polus.mock.horovod": ["init", "local_rank", "size", "DistributedGradientTape", "broadcast_variables", "allgather_object"],
    }.items():
        mod = importlib.import_module(name)
        for s in symbols:
            assert hasattr(mod, s), f"{name}.{s}"
    import polus.mock.horovod as hvd
    assert hvd.init() == "mock" and hvd.size() == 1 and hvd.local_rank() == 0 and hvd.allgather_object(3) == [3]


def test_core_and_utils_helpers():
    """tests/test_core.py:5-11 and tests/test_utils.py:26-55 of the reference."""
    import os
    from polus.core import find_dtype_and_shapes, get_jit_compile, set_jit_compile
    from polus.utils import complex_json_deserializer, complex_json_serializer, flatten_dict, merge_dicts
    import numpy as np
    os.environ.pop("POLUS_JIT", None)
    assert get_jit_compile() is False
    set_jit_compile(True)
    assert get_jit_compile() is True
    set_jit_compile(False)
    assert flatten_dict({"a": 1, "m": {"a": 2, "b": {"c": 3}}}) == {"a": 2, "c": 3}          # last occurrence wins
    assert merge_dicts({"a": 1}, {"a": 2, "b": 3}, {"b": 4}) == {"a": 2, "b": 4}
    cfg = {"model": {"w": np.arange(6, dtype=np.float32).reshape(2, 3), "n": 3}, "name": "x"}
    back = complex_json_deserializer(complex_json_serializer(cfg))
    assert back["name"] == "x" and back["model"]["n"] == 3 and back["model"]["w"].dtype == np.float32
    assert np.array_equal(back["model"]["w"], cfg["model"]["w"])
    dt, sh = find_dtype_and_shapes(({"ids": np.zeros((n, 4), np.int32), "y": 1.5} for n in (3, 5, 3)), k=3)
    assert sh == {"ids": (None, 4), "y": ()} and dt["ids"] == np.int32
