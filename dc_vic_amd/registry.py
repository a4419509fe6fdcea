"""name -> class registries, the plugin API the codec sits behind.

Mirror of src/utils/registry.py:11-92 (reference): `@X_REGISTRY.register()` keys a class by its
`__name__`, `X_REGISTRY.get(name)` returns it (KeyError when unknown).  The same registry objects
exist here under the same names so reference YAML `type:` strings resolve unchanged; the registries
the inference path never touches (trainer/optimizer/...) exist but stay empty.
"""
from __future__ import annotations

import inspect
import os.path as osp
from typing import Any, Dict, Optional


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map: Dict[str, Dict[str, Any]] = {}

    def _do_register(self, name: str, obj: Any, filename: str) -> None:
        if name in self._obj_map:
            raise AssertionError(f"An object named '{name}' was already registered in '{self._name}' registry!")
        self._obj_map[name] = {"obj": obj, "filename": filename}

    def register(self):
        def deco(func_or_class):
            filename = osp.basename(inspect.stack()[1].filename)
            self._do_register(func_or_class.__name__, func_or_class, filename)
            return func_or_class
        return deco

    def get(self, class_name: str, display_name: Optional[str] = None):
        ret = self._obj_map.get(class_name)
        if ret is None:
            raise KeyError(f"No object named '{class_name}' found in '{self._name}' registry!")
        return ret["obj"]

    def __contains__(self, name: str) -> bool:
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())

    def keys(self):
        return self._obj_map.keys()


TRAINER_REGISTRY = Registry("trainer")
OPTIMIZER_REGISTRY = Registry("optimizer")
SCHEDULER_REGISTRY = Registry("scheduler")

MODEL_REGISTRY = Registry("comp_model")
ENCODER_REGISTRY = Registry("encoder")
DECODER_REGISTRY = Registry("decoder")
HYPERENCODER_REGISTRY = Registry("hyperencoder")
HYPERDECODER_REGISTRY = Registry("hyperdecoder")
CONTEXTMODEL_REGISTRY = Registry("context_model")
ENTROPYMODEL_REGISTRY = Registry("entropy_model")
DISCRIMINATOR_REGISTRY = Registry("discriminator")
LRP_REGISTRY = Registry("residual_predictor")

DATASET_REGISTRY = Registry("dataset")
LOSS_REGISTRY = Registry("loss")
METRIC_REGISTRY = Registry("metric")

VQ_ESTIMATOR_REGISTRY = Registry("vq_estimator")
VQ_FUSION_REGISTRY = Registry("vq_fusion")
