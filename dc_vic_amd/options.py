"""YAML configuration with `_base_` inheritance and `_delete_` replacement.

Mirror of the behaviour of src/utils/options.py:63-130 (reference, mmcv-style): a config may name
one or several `_base_` files (paths relative to the child); bases are loaded recursively, must not
share top-level keys, and the child is deep-merged over them; a child dict carrying
`_delete_: True` replaces the base dict instead of merging.  Values are reachable as attributes
(`opt.subnet.encoder`) like the reference's addict-based ConfigDict; unknown attributes raise.
"""
from __future__ import annotations

import argparse
import copy
import os.path as osp
from typing import Dict, List, Tuple

import yaml

BASE_KEY = "_base_"
DELETE_KEY = "_delete_"


class ConfigDict(dict):
    """dict with attribute access; nested dicts are converted on the way in."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'{self.__class__.__name__}' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self) -> dict:
        def un(v):
            if isinstance(v, dict):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(un(x) for x in v)
            return v
        return un(self)


class BaseConfig:
    @staticmethod
    def _file2dict_yaml(filename: str) -> Tuple[Dict, str, List[str]]:
        filename = osp.abspath(osp.expanduser(filename))
        if not osp.isfile(filename):
            raise FileNotFoundError(f'file "{filename}" does not exist')
        if osp.splitext(filename)[1] != ".yaml":
            raise IOError("Only yaml type are supported now!")
        with open(filename, "r", encoding="utf-8") as f:
            text = f.read()
        cfg_dict = yaml.safe_load(text) or {}
        cfg_text = filename + "\n" + text
        loaded = [filename]
        if BASE_KEY in cfg_dict:
            bases = cfg_dict.pop(BASE_KEY)
            bases = bases if isinstance(bases, list) else [bases]
            merged: Dict = {}
            texts = []
            for b in bases:
                d, t, l = BaseConfig._file2dict_yaml(osp.join(osp.dirname(filename), b))
                dup = merged.keys() & d.keys()
                if dup:
                    raise KeyError(f"Duplicate key is not allowed among bases. Duplicate keys: {dup}")
                merged.update(d)
                texts.append(t)
                loaded.extend(l)
            cfg_dict = BaseConfig._merge_a_into_b(cfg_dict, merged)
            cfg_text = "\n".join(texts + [cfg_text])
        return cfg_dict, cfg_text, loaded

    @staticmethod
    def _merge_a_into_b(a: Dict, b: Dict) -> Dict:
        b = dict(b)
        for k, v in a.items():
            if isinstance(v, dict) and k in b and not v.pop(DELETE_KEY, False):
                if not isinstance(b[k], dict):
                    raise TypeError(f"{k}={v} in child config cannot inherit from base because {k} is a dict in the child "
                                    f"config but is of type {type(b[k])} in base config. You may set `{DELETE_KEY}=True` "
                                    "to ignore the base config")
                b[k] = BaseConfig._merge_a_into_b(v, b[k])
            else:
                if isinstance(v, dict):
                    v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
                b[k] = v
        return b

    @classmethod
    def fromfile(cls, filename: str, overrides: Dict = None) -> ConfigDict:
        cfg, text, _ = cls._file2dict_yaml(filename)
        if overrides:
            cfg = cls._merge_a_into_b(dict(overrides), cfg)
        out = ConfigDict(cfg)
        out["_cfg_text"] = text
        out["_filename"] = filename
        return out


def compress_arg_parser() -> argparse.ArgumentParser:
    """Flags of scripts/compress.py:39-50 (reference) + the multi-GPU additions of this build."""
    p = argparse.ArgumentParser()
    p.add_argument("--config_path", type=str, help="path to .yaml")
    p.add_argument("--model_path", type=str, help="path to model")
    p.add_argument("--img_dir", type=str)
    p.add_argument("--save_dir", type=str)
    p.add_argument("-q", "--quality", type=int, required=True)
    p.add_argument("--decompress", action="store_true")
    p.add_argument("-d", "--device", type=str, default="cuda:0")
    return p
