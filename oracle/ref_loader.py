"""Container-only loader for the REFERENCE model zoo — test infrastructure, never shipped.

``/root/reference/src/face_models.py`` imports ``torchvision.models`` (`face_models.py:7`), which is
not installed here and is not vendored by the reference.  This module registers a stub
``torchvision.models`` exposing ``resnet18(weights=...)`` / ``ResNet18_Weights.IMAGENET1K_V1`` (a
plain CPU ``nn.Module`` with torchvision-identical child names and order, so the reference's
``children()[:-1]`` / ``[:-2]`` slicing at `face_models.py:100,271,464,660` behaves) and then loads
the single reference file by path (never the ``src`` package: `src/__init__.py:9-23` pulls cv2 /
facenet_pytorch / albumentations, `base_config.py:39-42` mkdirs under the read-only tree).

Only ``oracle/gen_golden.py`` and the container-only tests use it.  ``/root/reference`` does not
exist on the GPU box; ``available()`` says whether it can be used.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF_FILE = "/root/reference/src/face_models.py"


class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idn = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            idn = self.downsample(x)
        return self.relu(out + idn)


class _ResNet18(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inpl = 64
        for i, (planes, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), start=1):
            ds = None
            if stride != 1 or inpl != planes:
                ds = nn.Sequential(nn.Conv2d(inpl, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
            setattr(self, f"layer{i}", nn.Sequential(_BasicBlock(inpl, planes, stride, ds),
                                                      _BasicBlock(planes, planes)))
            inpl = planes
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def _install_stub() -> None:
    tv = types.ModuleType("torchvision")
    tv.__frmap_stub__ = True
    tvm = types.ModuleType("torchvision.models")

    class ResNet18_Weights:  # sentinel only; pretrained weights are a remote fetch (not attempted)
        IMAGENET1K_V1 = "IMAGENET1K_V1"

    def resnet18(weights=None, **kw):
        return _ResNet18()

    tvm.resnet18 = resnet18
    tvm.ResNet18_Weights = ResNet18_Weights
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm


def available() -> bool:
    return os.path.isfile(REF_FILE)


_cached = None


def load_reference():
    """Return the reference ``face_models`` module (executing the reference's own class code)."""
    global _cached
    if _cached is not None:
        return _cached
    if not available():
        raise FileNotFoundError(REF_FILE)
    try:
        import torchvision  # noqa: F401
    except Exception:
        _install_stub()
    spec = importlib.util.spec_from_file_location("ref_face_models", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode, old = True, sys.dont_write_bytecode
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = old
    _cached = mod
    return mod


APP_FILE = "/root/reference/src/app.py"


def load_reference_function(path: str, name: str, namespace: dict):
    """Return the reference's own top-level function ``name`` of ``path``, compiled from its source in the build
    container WITHOUT importing the module around it (``src/app.py`` imports streamlit / cv2 / facenet_pytorch at
    module level, none of which exist here): the file is parsed, the one ``FunctionDef`` node is compiled on its own
    and executed in ``namespace`` (the globals it needs, e.g. ``{"torch": torch}``).  Nothing is written anywhere;
    the reference's text never leaves the container."""
    import ast
    with open(path, "r", encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            ns = dict(namespace)
            exec(compile(mod, path, "exec"), ns)
            return ns[name]
    raise LookupError(f"{path} has no top-level function {name!r}")


def reference_compare_faces():
    """`src/app.py:50-64` itself (needs only ``torch``)."""
    return load_reference_function(APP_FILE, "compare_faces", {"torch": torch})
