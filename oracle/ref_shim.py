"""TEST INFRASTRUCTURE (container-only): import the read-only reference at /root/reference.

This module is used ONLY by ``oracle/make_golden.py`` (to generate the committed fixtures under
``tests/golden/``) and by ``tests/test_oracle_vs_reference.py`` (skipped when /root/reference is absent,
i.e. on the GPU box).  Nothing in the product path (``climategan_amd/``), in ``bench.py`` or in
``__graft_entry__.smoke()`` imports it: the reference never travels.

Recipe:
  * the reference's ``climategan/__init__.py:4-8`` imports every submodule (and thus comet_ml, kornia,
    skimage ... which are not installed) -> register a synthetic package whose ``__path__`` points at the
    reference directory so ``__init__`` is never executed;
  * ``addict`` is replaced by a small auto-vivifying dict; other missing third-party modules by
    attribute-generating dummies whose attributes are classes.
"""
import importlib
import importlib.machinery
import sys
import types
from pathlib import Path

REF_ROOT = Path("/root/reference")


def available() -> bool:
    return (REF_ROOT / "climategan" / "painter.py").exists()


class Dict(dict):
    """Minimal stand-in for addict.Dict (auto-vivifying attribute dict)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        src = dict(*args, **kwargs)
        for k, v in src.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(i) for i in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        if k not in self:
            self[k] = Dict()
        return self[k]

    def __setattr__(self, k, v):
        self[k] = v

    def __missing__(self, k):
        v = Dict()
        super().__setitem__(k, v)
        return v

    def copy(self):
        return Dict(self)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def to_dict(self):
        out = {}
        for k, v in self.items():
            out[k] = v.to_dict() if isinstance(v, Dict) else v
        return out


class _Dummy(types.ModuleType):
    """Module whose every missing attribute is a fresh class (so it can be subclassed)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def _dummy(name):
    if name in sys.modules:
        return sys.modules[name]
    m = _Dummy(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        setattr(_dummy(parent), child, m)
    return m


_DUMMIES = [
    "comet_ml", "skimage", "skimage.io", "skimage.color", "skimage.transform", "skimage.filters",
    "torchvision", "torchvision.models", "torchvision.models.inception", "torchvision.transforms",
    "torchvision.transforms.functional", "torchvision.utils", "kornia", "kornia.filters",
    "kornia.filters.kernels", "imageio", "cv2", "sklearn", "sklearn.metrics", "sklearn.metrics.pairwise",
    "matplotlib", "matplotlib.pyplot", "seaborn", "torch_optimizer", "hydra", "omegaconf",
]

_installed = False


def install():
    """Idempotently install the stubs and the synthetic ``climategan`` package."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present at /root/reference (expected on the GPU box)")
    addict = types.ModuleType("addict")
    addict.Dict = Dict
    sys.modules["addict"] = addict
    for n in _DUMMIES:
        try:
            importlib.import_module(n)
        except Exception:
            _dummy(n)
    pkg = types.ModuleType("climategan")
    pkg.__path__ = [str(REF_ROOT / "climategan")]
    pkg.__spec__ = importlib.machinery.ModuleSpec("climategan", None, is_package=True)
    pkg.__spec__.submodule_search_locations = pkg.__path__
    sys.modules["climategan"] = pkg
    _installed = True


def ref(module: str):
    """Import ``climategan.<module>`` from the reference tree."""
    install()
    return importlib.import_module("climategan." + module)


def default_opts():
    """opts = defaults.yaml + events.yaml, built without ``load_opts`` (which asserts data files)."""
    import yaml

    install()
    opts = Dict(yaml.safe_load((REF_ROOT / "shared/trainer/defaults.yaml").read_text()))
    opts.events = Dict(yaml.safe_load((REF_ROOT / "shared/trainer/events.yaml").read_text()))
    opts.val.val_painter = "none"
    return opts
