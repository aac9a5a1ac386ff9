"""Drop-in aliasing: make ``import pointmvsnet.<operator module>`` resolve to this package.

The drop-in boundary is the import list of the reference model graph (reference model.py:8-12,
SURVEY.md section 8(b)).  After ``install_as_pointmvsnet()`` the reference's *unmodified* ``model.py``
can be loaded with ``load_reference_model(path)`` and runs on the HIP operators.
"""
import importlib
import importlib.util
import sys
import types

_ALIASES = {
    "pointmvsnet.networks": "pointmvsnet_amd.networks",
    "pointmvsnet.functions": "pointmvsnet_amd.functions",
    "pointmvsnet.functions.functions": "pointmvsnet_amd.functions.functions",
    "pointmvsnet.functions.gather_knn": "pointmvsnet_amd.functions.gather_knn",
    "pointmvsnet.utils": "pointmvsnet_amd.utils",
    "pointmvsnet.utils.feature_fetcher": "pointmvsnet_amd.utils.feature_fetcher",
    "pointmvsnet.utils.torch_utils": "pointmvsnet_amd.utils.torch_utils",
    "pointmvsnet.nn": "pointmvsnet_amd.nn",
    "pointmvsnet.nn.conv": "pointmvsnet_amd.nn.conv",
    "pointmvsnet.nn.mlp": "pointmvsnet_amd.nn.mlp",
    "pointmvsnet.nn.init": "pointmvsnet_amd.nn.init",
}


def install_as_pointmvsnet():
    """Register this package's operator modules under the reference's module names."""
    pkg = sys.modules.get("pointmvsnet")
    if pkg is None or getattr(pkg, "__pointflow_amd__", False) is False:
        pkg = types.ModuleType("pointmvsnet")
        pkg.__path__ = []
        pkg.__pointflow_amd__ = True
        sys.modules["pointmvsnet"] = pkg
    for alias, target in _ALIASES.items():
        mod = importlib.import_module(target)
        sys.modules[alias] = mod
        parent, _, leaf = alias.rpartition(".")
        setattr(sys.modules[parent], leaf, mod)
    # the drop-in route's two switches (round 6; PF_DROPIN_FAST=0 keeps round 5's behaviour): the operator modules'
    # inference forwards replay from per-module hipGraphs (graph.module_forward), and get_pixel_grids hands model.py a
    # device tensor so that its ``.to(device)`` is no synchronous host-to-device copy (functions/functions.py)
    import os
    import torch
    if os.environ.get("PF_DROPIN_FAST", "1") != "0" and torch.cuda.is_available():
        from . import graph
        from .functions import functions
        graph.MODULE_GRAPHS = True
        functions.PIXEL_GRID_ON_DEVICE = True
    return pkg


def load_reference_model(model_py_path):
    """Execute a reference ``pointmvsnet/model.py`` file, unmodified, on top of the aliased operators."""
    install_as_pointmvsnet()
    import importlib.machinery
    loader = importlib.machinery.SourceFileLoader("pointmvsnet.model", model_py_path)   # any file name
    spec = importlib.util.spec_from_loader("pointmvsnet.model", loader)
    module = importlib.util.module_from_spec(spec)
    sys.modules["pointmvsnet.model"] = module
    spec.loader.exec_module(module)
    return module
