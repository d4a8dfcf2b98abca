"""Import harness for the *reference* implementation (test infrastructure only).

TEST INFRASTRUCTURE -- never imported by the product path.  Only
``oracle/make_golden.py`` and ``tests/test_oracle_vs_reference.py`` use it, and only in
the build container where ``/root/reference`` exists (it does not exist on the GPU box).

The reference's hot path is plain Python but cannot be imported
with ``import src.models`` because ``src/models/__init__.py`` pulls in cv2 / diffusers /
torchvision.  This harness registers *bare namespace packages* for ``src`` and
``src.models`` so that only the files on the hot path are executed, and pre-seeds
``sys.modules`` with three thin stand-ins for pip packages that are absent here:

* ``timm.models.layers``  -> ``to_2tuple``, ``trunc_normal_`` (= torch's), identity ``DropPath``
* ``omegaconf.OmegaConf.create`` -> ``types.SimpleNamespace``
* ``mamba_ssm.ops.selective_scan_interface`` -> ``selective_scan_fn`` / ``selective_scan_ref``
  bound to OUR restatement of the published recurrence (oracle/ffsr_oracle/scan.py).
  mamba-ssm 2.3.0 (scripts/kaggle_inference_fixed.py:19) is an un-vendored CUDA wheel,
  so the scan arithmetic is "parity unpinned" (no reference test or fixture pins it).

Nothing from the reference is copied: the modules are imported from where they lie.
"""
from __future__ import annotations

import collections.abc
import importlib
import itertools
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("FFSR_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "models"))


def _install_shims():
    import torch
    import torch.nn as nn

    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only tree

    # --- bare namespace packages (skip the reference's heavy __init__.py files) ---
    for name, rel in (("src", "src"), ("src.models", "src/models")):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__path__ = [os.path.join(REFERENCE_ROOT, rel)]
            sys.modules[name] = mod
    sys.modules["src"].models = sys.modules["src.models"]

    # --- timm.models.layers ---
    if "timm" not in sys.modules:
        def _ntuple(n):
            def parse(x):
                if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
                    return tuple(x)
                return tuple(itertools.repeat(x, n))
            return parse

        class DropPath(nn.Module):  # eval-mode identity (drop_path only acts in training)
            def __init__(self, drop_prob=0.0, scale_by_keep=True):
                super().__init__()
                self.drop_prob = drop_prob

            def forward(self, x):
                return x

        timm = types.ModuleType("timm")
        timm_models = types.ModuleType("timm.models")
        timm_layers = types.ModuleType("timm.models.layers")
        timm_layers.to_2tuple = _ntuple(2)
        timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
        timm_layers.DropPath = DropPath
        timm.models = timm_models
        timm_models.layers = timm_layers
        timm.layers = timm_layers
        sys.modules.update({"timm": timm, "timm.models": timm_models,
                            "timm.models.layers": timm_layers, "timm.layers": timm_layers})

    # --- omegaconf ---
    if "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")

        class OmegaConf:  # noqa: D401 - stand-in
            @staticmethod
            def create(d):
                return types.SimpleNamespace(**d)

        oc.OmegaConf = OmegaConf
        sys.modules["omegaconf"] = oc

    # --- mamba_ssm (scan = our restatement; parity unpinned) ---
    if "mamba_ssm" not in sys.modules:
        here = os.path.dirname(os.path.abspath(__file__))
        if here not in sys.path:
            sys.path.insert(0, here)
        from ffsr_oracle.scan import selective_scan_ref as _scan

        ms = types.ModuleType("mamba_ssm")
        ops = types.ModuleType("mamba_ssm.ops")
        ssi = types.ModuleType("mamba_ssm.ops.selective_scan_interface")
        ssi.selective_scan_fn = _scan
        ssi.selective_scan_ref = _scan
        ms.ops = ops
        ops.selective_scan_interface = ssi
        sys.modules.update({"mamba_ssm": ms, "mamba_ssm.ops": ops,
                            "mamba_ssm.ops.selective_scan_interface": ssi})


_CACHE = {}


def load_reference():
    """Returns a namespace with the reference classes / factories of the hot path."""
    if "ns" in _CACHE:
        return _CACHE["ns"]
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    _install_shims()
    import contextlib
    import io

    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        fusion = importlib.import_module("src.models.enhanced_fusion_v2")
        nafnet = importlib.import_module("src.models.nafnet")
        drct = importlib.import_module("src.models.drct")
        grl = importlib.import_module("src.models.grl")
        mamba = importlib.import_module("src.models.mambair.mambair_arch")
    ns = types.SimpleNamespace(
        CompleteEnhancedFusionSR=fusion.CompleteEnhancedFusionSR,
        fusion_module=fusion,
        create_nafnet_sr_model=nafnet.create_nafnet_sr_model,
        NAFNetSR=nafnet.NAFNetSR,
        create_drct_model=drct.create_drct_model,
        DRCT=drct.DRCT,
        create_grl_model=grl.create_grl_model,
        GRL=grl.GRL,
        MambaIR=mamba.MambaIR,
        mamba_module=mamba,
    )
    _CACHE["ns"] = ns
    return ns
