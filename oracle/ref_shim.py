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


def install_torchaudio_transforms():
    """FreqCodec builds `torchaudio.transforms.Spectrogram / InverseSpectrogram` (codec_freq.py:185-210).  torchaudio is a stub
    in this image, so the two transforms are provided here exactly as torchaudio publishes them: thin wrappers over `torch.stft`
    / `torch.istft` (center=True, pad_mode="reflect", periodic Hann window of win_length = n_fft, normalized=False,
    onesided=True).  The FreqCodec goldens are therefore pinned to the reference's own code over THIS restatement of the
    third-party transform (said so in tests/golden/MANIFEST.json)."""
    import torch
    import torchaudio

    class Spectrogram(torch.nn.Module):
        def __init__(self, n_fft=400, win_length=None, hop_length=None, pad=0, power=2.0, normalized=False, center=True,
                     pad_mode="reflect", onesided=True, **_):
            super().__init__()
            self.n_fft, self.win_length = n_fft, win_length or n_fft
            self.hop_length, self.power = hop_length or self.win_length // 2, power
            self.register_buffer("window", torch.hann_window(self.win_length), persistent=False)

        def forward(self, x):
            shape = x.shape
            s = torch.stft(x.reshape(-1, shape[-1]), self.n_fft, self.hop_length, self.win_length, self.window, center=True,
                           pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
            s = s.reshape(shape[:-1] + s.shape[-2:])
            if self.power is None:
                return s
            return s.abs() if self.power == 1 else s.abs().pow(self.power)

    class InverseSpectrogram(torch.nn.Module):
        def __init__(self, n_fft=400, win_length=None, hop_length=None, pad=0, normalized=False, center=True,
                     pad_mode="reflect", onesided=True, **_):
            super().__init__()
            self.n_fft, self.win_length = n_fft, win_length or n_fft
            self.hop_length = hop_length or self.win_length // 2
            self.register_buffer("window", torch.hann_window(self.win_length), persistent=False)

        def forward(self, s, length=None):
            shape = s.shape
            w = torch.istft(s.reshape(-1, shape[-2], shape[-1]), self.n_fft, self.hop_length, self.win_length, self.window,
                            center=True, normalized=False, onesided=True, length=length, return_complex=False)
            return w.reshape(shape[:-2] + w.shape[-1:])

    torchaudio.transforms.Spectrogram = Spectrogram
    torchaudio.transforms.InverseSpectrogram = InverseSpectrogram


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
