"""polus/utils.py drop-in: the helpers the model (de)serialisers and user scripts use, with
torch / NumPy arrays where the reference has tf tensors."""
import json
import random

import numpy as np


def set_random_seed(seed_value=42):
    """polus/utils.py:6-9 (tf.random.set_seed -> torch.manual_seed)."""
    import torch
    torch.manual_seed(seed_value)
    random.seed(seed_value)
    np.random.seed(seed_value)


def merge_dicts(*list_of_dicts):
    """polus/utils.py:11-19: later dicts win."""
    temp = dict(list_of_dicts[0], **list_of_dicts[1])
    for i in range(2, len(list_of_dicts)):
        temp.update(list_of_dicts[i])
    return temp


def flatten_dict(d):
    """polus/utils.py:21-35: nested dicts flattened onto their leaf keys; a duplicated key keeps the LAST
    occurrence (tests/test_utils.py:26-55)."""
    items = []
    for k, v in d.items():
        if isinstance(v, dict):
            items.extend(flatten_dict(v).items())
        else:
            items.append((k, v))
    return dict(items)


def unique(iterable, key=lambda x: x):
    return list({key(x): x for x in iterable}.values())


def is_jsonable(x):
    try:
        json.dumps(x)
        return True
    except (TypeError, OverflowError):
        return False


def _is_tensor(v):
    try:
        import torch
        if isinstance(v, torch.Tensor):
            return True
    except ImportError:
        pass
    return isinstance(v, np.ndarray)


def complex_json_serializer(data):
    """polus/utils.py:52-65: tensors become {"_class": "tensor", "dtype", "values"}."""
    _dict = {}
    for k, v in data.items():
        if isinstance(v, dict):
            _dict[k] = complex_json_serializer(v)
        elif is_jsonable(v):
            _dict[k] = v
        elif _is_tensor(v):
            a = v.detach().cpu().numpy() if not isinstance(v, np.ndarray) else v
            _dict[k] = {"_class": "tensor", "dtype": str(a.dtype), "values": a.tolist()}
        else:
            raise ValueError(f"Cannot serialize {type(v)} please add a json serializer to this type of data")
    return _dict


def complex_json_deserializer(data):
    """polus/utils.py:67-81 (tensors come back as NumPy arrays of the recorded dtype)."""
    _dict = {}
    for k, v in data.items():
        if isinstance(v, dict):
            if "_class" not in v:
                _dict[k] = complex_json_deserializer(v)
            elif v["_class"] == "tensor":
                _dict[k] = np.asarray(v["values"], dtype=v["dtype"])
            else:
                _type = v["_class"]
                raise ValueError(f"Cannot deserialize {_type} please add a json deserializer to this type of data")
        else:
            _dict[k] = v
    return _dict


class Singleton(type):
    """polus/utils.py:84-96."""

    def __init__(self, *args, **kwargs):
        self.__instance = None
        super().__init__(*args, **kwargs)

    def __call__(self, *args, **kwargs):
        if self.__instance is None:
            self.__instance = super().__call__(*args, **kwargs)
        return self.__instance
