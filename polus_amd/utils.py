"""Helper surface of polus/utils.py, written for this engine: seeding, dictionary plumbing for the model
configuration files (`.cfg` = JSON) and the process-wide singleton metaclass.

Same names and observable behaviour as the reference (the `.cfg` files must stay readable by either side); arrays
are NumPy / torch where the reference has tf tensors.  What each helper must do is pinned by the reference's
tests/test_utils.py (singleton identity, last-key-wins flattening, JSON round trip of array leaves)."""
import json
import random

import numpy as np

_TENSOR_TAG = "tensor"


def set_random_seed(seed_value=42):
    """One seed for every generator a training script may draw from (polus/utils.py:6-9 seeds tf, random, numpy)."""
    import torch
    random.seed(seed_value)
    np.random.seed(seed_value)
    torch.manual_seed(seed_value)


def merge_dicts(*list_of_dicts):
    """Left-to-right union: a key present in several dictionaries takes the value of the right-most one
    (polus/utils.py:11-19).  The inputs are left untouched."""
    merged = {}
    for d in list_of_dicts:
        merged.update(d)
    return merged


def _walk_leaves(tree):
    """(key, value) of every non-dict entry, depth first, in insertion order."""
    for key, value in tree.items():
        if isinstance(value, dict):
            yield from _walk_leaves(value)
        else:
            yield key, value


def flatten_dict(d):
    """Drops the nesting of a configuration: every leaf lands under its own key; when the same key occurs twice the
    occurrence met LAST in a depth-first walk wins (polus/utils.py:21-35, tests/test_utils.py:26-55)."""
    flat = {}
    for key, value in _walk_leaves(d):
        flat[key] = value
    return flat


def unique(iterable, key=lambda x: x):
    """One element per distinct `key(element)`: the last one seen, in order of first appearance of its key."""
    by_key = {}
    for element in iterable:
        by_key[key(element)] = element
    return list(by_key.values())


def is_jsonable(x):
    """Whether json.dumps accepts `x` as it stands."""
    try:
        json.dumps(x)
    except (TypeError, OverflowError):
        return False
    return True


def _as_array(value):
    """NumPy view of an array-like leaf (torch tensor on any device, NumPy array), or None."""
    if isinstance(value, np.ndarray):
        return value
    if hasattr(value, "detach") and hasattr(value, "cpu"):
        return value.detach().cpu().numpy()
    return None


def complex_json_serializer(data):
    """JSON-ready copy of a (nested) configuration.  Array leaves are written as
    {"_class": "tensor", "dtype": ..., "values": nested lists} -- the record layout of polus/utils.py:52-65, so a `.cfg`
    written here reads back there and vice versa.  Anything else json cannot take is an error, not a silent drop."""
    out = {}
    for key, value in data.items():
        if isinstance(value, dict):
            out[key] = complex_json_serializer(value)
            continue
        if is_jsonable(value):
            out[key] = value
            continue
        arr = _as_array(value)
        if arr is None:
            raise ValueError(f"Cannot serialize {type(value)} please add a json serializer to this type of data")
        out[key] = {"_class": _TENSOR_TAG, "dtype": str(arr.dtype), "values": arr.tolist()}
    return out


def complex_json_deserializer(data):
    """Inverse of complex_json_serializer (polus/utils.py:67-81): tagged records come back as NumPy arrays of the
    recorded dtype, plain dictionaries are descended into, unknown tags are refused."""
    out = {}
    for key, value in data.items():
        if not isinstance(value, dict):
            out[key] = value
        elif "_class" not in value:
            out[key] = complex_json_deserializer(value)
        elif value["_class"] == _TENSOR_TAG:
            out[key] = np.asarray(value["values"], dtype=value["dtype"])
        else:
            raise ValueError(f"Cannot deserialize {value['_class']} please add a json deserializer to this type of data")
    return out


from .context import Singleton  # noqa: E402,F401  (polus/utils.py:84-96 exports it; PolusContext and HPOContext use it)
