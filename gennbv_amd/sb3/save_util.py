"""SB3-zip checkpoints: read what the reference's `PPO_Grid_Obs.save()` writes and write
archives its `set_parameters()` / `load_from_zip_file()` can read.

Layout (stable_baselines3/common/save_util.py:289-330 of the reference): a ZIP_STORED archive with
  data                         JSON of the algorithm's __dict__ minus the excluded attributes; values that
                               are not JSON-serialisable are {":type:", ":serialized:" = base64(cloudpickle)}
                               plus their first-level JSON-able attributes (save_util.py:74-126)
  policy.pth                   torch.save(policy.state_dict())
  policy.optimizer.pth         torch.save(policy.optimizer.state_dict())      (torch.optim.Adam format)
  pytorch_variables.pth        torch.save({})  (on-policy algorithms have none)
  _stable_baselines3_version   text
  system_info.txt              text

Reading never needs the reference's classes: the two state dicts are plain tensors (`weights_only=True`) and from
`data` only JSON-plain hyper-parameters are consumed.  ":serialized:" blobs go through a RESTRICTED unpickler
(`_SafeUnpickler`: plain containers / numbers, numpy arrays and dtypes, torch dtypes, and an explicit list of this package's data classes)
-- anything else (the reference's own classes, cloudpickled lambdas, arbitrary callables) is skipped and listed in
the returned `skipped`, so loading an untrusted archive cannot execute code from it.  `trusted=True` restores SB3's
behaviour (full cloudpickle) for archives you wrote yourself.
"""
from __future__ import annotations

import base64
import io
import json
import os
import pickle
import platform
import zipfile
from typing import Any, Dict, Optional, Tuple

import torch

SB3_VERSION = "1.6.0"  # the reference's vendored stable_baselines3/version.txt

# attributes of the algorithm object that never go into `data` (base_class_grid_obs.py:313-333 + the
# torch-saved ones, :826-833) -- plus this build's device-side state
EXCLUDED = {"policy", "device", "env", "eval_env", "replay_buffer", "rollout_buffer", "_vec_normalize_env", "_episode_storage",
            "_logger", "_custom_logger", "_hip", "_sync", "_pending", "_last_obs", "_last_episode_starts", "ep_info_buffer"}


def _jsonable(x) -> bool:
    try:
        json.dumps(x)
        return True
    except (TypeError, OverflowError, ValueError):
        return False


def data_to_json(data: Dict[str, Any]) -> str:
    """save_util.py:74-126 of the reference."""
    try:
        import cloudpickle as pk
    except ImportError:  # pragma: no cover
        pk = pickle
    out = {}
    for key, item in data.items():
        if _jsonable(item):
            out[key] = item
            continue
        entry = {":type:": str(type(item)), ":serialized:": base64.b64encode(pk.dumps(item)).decode()}
        if hasattr(item, "__dict__") or isinstance(item, dict):
            for k, v in (item.items() if isinstance(item, dict) else item.__dict__.items()):
                entry[str(k)] = v if _jsonable(v) else str(v)
        out[key] = entry
    return json.dumps(out, indent=4)


_SAFE_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray", "complex",
                  "slice", "range", "NoneType"}
_SAFE_GLOBALS = {("collections", "OrderedDict"), ("collections", "deque"),
                 ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy", "float32"), ("numpy", "float64"), ("numpy", "int64"),
                 ("numpy", "int32"), ("numpy", "bool_"),
                 ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
                 ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
                 ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
                 ("torch", "float32"), ("torch", "float64"), ("torch", "bfloat16"), ("torch", "float16"), ("torch", "Size")}


# classes of this package that may appear in `data` (spaces, policy / extractor classes named by `policy_kwargs`, configs).
# An explicit (module, name) list: a module-prefix rule would let protocol-4 dotted names walk from any module of the
# package to whatever it imports (("gennbv_amd.sb3.save_util", "os.system") resolves to os.system).
_SAFE_PACKAGE = {("gennbv_amd.spaces", "Box"), ("gennbv_amd.spaces", "MultiDiscrete"),
                 ("gennbv_amd.network.hybrid_encoder", "Hybrid_Encoder"),
                 ("gennbv_amd.sb3.policies", "ActorCriticPolicy_Train_Eval"),
                 ("gennbv_amd.sb3.distributions", "MultiCategoricalDistribution"),
                 ("gennbv_amd.env.config", "TaskConfig"), ("gennbv_amd.env.config", "PPOConfig")}


class _SafeUnpickler(pickle.Unpickler):
    """Resolves only data-like globals and the package's own data classes, each by exact (module, name)."""

    def find_class(self, module, name):
        if "." in name:  # protocol >= 4 resolves dotted names through attributes: never needed for the allow-list
            raise pickle.UnpicklingError(f"dotted global {module}.{name} is not allowed")
        if module == "builtins" and name in _SAFE_BUILTINS:
            return super().find_class(module, name)
        if (module, name) in _SAFE_GLOBALS:
            return super().find_class(module, name)
        if (module, name) in _SAFE_PACKAGE:
            obj = super().find_class(module, name)
            if not (isinstance(obj, type) and obj.__module__ == module):
                raise pickle.UnpicklingError(f"{module}.{name} is not a class defined in {module}")
            return obj
        raise pickle.UnpicklingError(f"global {module}.{name} is not on the allow-list")


def _loads(blob: bytes, trusted: bool):
    if trusted:
        try:
            import cloudpickle as pk
        except ImportError:  # pragma: no cover
            pk = pickle
        return pk.loads(blob)
    return _SafeUnpickler(io.BytesIO(blob)).load()


def json_to_data(text: str, custom_objects: Optional[Dict[str, Any]] = None, trusted: bool = False) -> Tuple[Dict[str, Any], list]:
    """save_util.py:129-172; entries that cannot be (or may not be) unpickled here are skipped (returned by name)."""
    data, skipped = {}, []
    for key, item in json.loads(text).items():
        if custom_objects is not None and key in custom_objects:
            data[key] = custom_objects[key]
        elif isinstance(item, dict) and ":serialized:" in item:
            try:
                data[key] = _loads(base64.b64decode(item[":serialized:"].encode()), trusted)
            except Exception:  # class not importable here (the reference's own modules)
                skipped.append(key)
        else:
            data[key] = item
    return data, skipped


def save_to_zip_file(path, data: Optional[Dict[str, Any]], params: Dict[str, Any], pytorch_variables: Optional[Dict[str, Any]] = None) -> None:
    if isinstance(path, (str, os.PathLike)):
        path = os.fspath(path)
        if not path.endswith(".zip"):
            path += ".zip"
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with zipfile.ZipFile(path, mode="w") as z:
        if data is not None:
            z.writestr("data", data_to_json(data))
        with z.open("pytorch_variables.pth", mode="w", force_zip64=True) as f:
            torch.save(pytorch_variables if pytorch_variables is not None else {}, f)
        for name, sd in params.items():
            with z.open(name + ".pth", mode="w", force_zip64=True) as f:
                torch.save(sd, f)
        z.writestr("_stable_baselines3_version", SB3_VERSION)
        z.writestr("system_info.txt", f"OS: {platform.platform()}\nPython: {platform.python_version()}\nPyTorch: {torch.__version__}\n"
                                      f"GPU Enabled: {torch.cuda.is_available()}\nwritten by: gennbv_amd\n")


def load_from_zip_file(path, load_data: bool = True, custom_objects=None, device="cpu", trusted: bool = False):
    """-> (data or None, {name: state_dict}, pytorch_variables or None, skipped data keys)."""
    if isinstance(path, (str, os.PathLike)):
        path = os.fspath(path)
        if not os.path.exists(path) and os.path.exists(path + ".zip"):
            path += ".zip"
    data, skipped, params, variables = None, [], {}, None
    try:
        with zipfile.ZipFile(path) as z:
            names = z.namelist()
            if "data" in names and load_data:
                data, skipped = json_to_data(z.read("data").decode(), custom_objects, trusted)
            for n in names:
                if os.path.splitext(n)[1] != ".pth":
                    continue
                obj = torch.load(io.BytesIO(z.read(n)), map_location=device, weights_only=True)  # state dicts: tensors only
                if n in ("pytorch_variables.pth", "tensors.pth"):
                    variables = obj
                else:
                    params[os.path.splitext(n)[0]] = obj
    except zipfile.BadZipFile as e:
        raise ValueError(f"Error: the file {path} wasn't a zip-file") from e
    return data, params, variables, skipped
