"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Import shim that lets the *real* reference (`/root/reference`, read-only, only present in the
build container, never on the GPU box) run on CPU here so that golden fixtures can be generated
from it (`oracle/make_golden.py`) and the restated oracle (`oracle/torch_oracle.py`,
`oracle/c/`) can be pinned against it.

The reference needs 10 third-party packages that are absent from this image:
typeguard, torchaudio, librosa, kaldiio, thop, humanfriendly, soundfile, h5py, torch_complex,
pytorch_wpe.  None of them is on the arithmetic path; they are replaced by MagicMock modules.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = os.environ.get("FUNCODEC_REFERENCE", "/root/reference")

_STUBS = ("typeguard", "torchaudio", "librosa", "kaldiio", "thop", "humanfriendly",
          "soundfile", "h5py", "torch_complex", "pytorch_wpe")


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUBS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        m.__loader__ = self
        if spec.name == "typeguard":
            m.check_argument_types = lambda *a, **k: True
            m.check_return_type = lambda *a, **k: True
        return m

    def exec_module(self, module):
        return None


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "funcodec"))


def install():
    """Make `import funcodec...` resolve to the reference, with the missing packages stubbed."""
    if not available():
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
