import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dau-convnet_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests only make sense where a device is visible; everywhere else they are skipped
    # unless explicitly selected (then they fail loudly inside the test).
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# Modules that test the EXACT gather kernels (k_gather_mfma.hip: variants, patch / stack / window forms, bit-for-bit properties).
# By default a plan whose unit count makes them pay also holds the two-limb f16 dense members (k_dense_split.hip), which take
# over the gather-sum passes of calls with small offsets -- these modules would then test those instead of what they name.  Their
# plans are therefore created with DAU_FLAG_NO_DENSE_SPLIT; the dense members have their own parity tests
# (test_gpu_dense_split.py, the split sweep of test_gpu_fuzz.py) and run by default in the layer / C-ABI-host / smoke tests.
_EXACT_GATHER_MODULES = {"test_gpu_parity", "test_gpu_baseline_configs", "test_gpu_config_depth", "test_gpu_fullsize",
                         "test_gpu_bf16", "test_gpu_fuzz"}


def _force_exact(module):
    Plan = module.Plan
    if getattr(Plan, "_exact_patch", None):
        return Plan.__init__, None
    orig = Plan.__init__
    skip = module.FLAG_DENSE_SPLIT_F16 | module.FLAG_DENSE_BF16 | module.FLAG_NO_DENSE_SPLIT

    def init(self, *a, **kw):
        flags = kw.get("flags", module.FLAG_USE_INTERPOLATION)
        if not (int(flags) & skip):
            kw["flags"] = int(flags) | module.FLAG_NO_DENSE_SPLIT
        orig(self, *a, **kw)
    Plan.__init__ = init
    Plan._exact_patch = True
    return orig, Plan


@pytest.fixture(autouse=True)
def _exact_gather_modules(request):
    if request.module.__name__ not in _EXACT_GATHER_MODULES or "gpu" not in request.keywords:
        yield
        return
    undo = []
    try:
        from dau_conv import _capi
        import util
        mods = [_capi] + [m for m in util._TUNING]
        if not util._TUNING:
            try:
                mods.append(util.tuning_capi())
            except Exception:
                pass
        for m in mods:
            orig, Plan = _force_exact(m)
            if Plan is not None:
                undo.append((Plan, orig))
    except ImportError:
        pass
    yield
    for Plan, orig in undo:
        Plan.__init__ = orig
        Plan._exact_patch = None
