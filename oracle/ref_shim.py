"""Container-only loader for the reference's own Python modules.

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_golden.py (and by the optional
``-m "not gpu"`` cross-checks when /root/reference is present) to run the
reference's *own* ``util/networks.py`` / ``util/meshnet.py`` / ``util/mesh.py``
unchanged, from where they lie under /root/reference.  Nothing is copied; on
the GPU box /root/reference does not exist and ``available()`` returns False.

Two shims are needed:
  * ``turtle``          -- util/mesh.py:1 does ``from turtle import pd`` (tkinter absent)
  * ``torch_geometric`` -- absent third-party dependency; replaced by
                           oracle.pyg_restatement (ChebConv, GCNConv, Sequential, Data)
"""
from __future__ import annotations

import importlib
import os
import sys
import types
import warnings

REFERENCE_ROOT = os.environ.get("SEMIGCN_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "util", "networks.py"))


def _install_shims():
    from . import pyg_restatement as P

    if "turtle" not in sys.modules:
        t = types.ModuleType("turtle")
        t.pd = None
        sys.modules["turtle"] = t
    if "torch_geometric" not in sys.modules:
        tg = types.ModuleType("torch_geometric")
        tg_nn = types.ModuleType("torch_geometric.nn")
        tg_data = types.ModuleType("torch_geometric.data")
        tg_nn.ChebConv, tg_nn.GCNConv, tg_nn.Sequential = P.ChebConv, P.GCNConv, P.Sequential
        tg_data.Data = P.Data
        tg.nn, tg.data = tg_nn, tg_data
        tg.__oracle_shim__ = True
        sys.modules["torch_geometric"] = tg
        sys.modules["torch_geometric.nn"] = tg_nn
        sys.modules["torch_geometric.data"] = tg_data


def load():
    """Return a namespace with the reference's util modules:
    ``.mesh .networks .meshnet .loss .models .datamaker``."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    _install_shims()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ns = types.SimpleNamespace()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name in ("mesh", "networks", "meshnet", "loss", "models", "datamaker"):
            setattr(ns, name, importlib.import_module(f"util.{name}"))
    return ns
