"""Backends for the kernel tests: 'emu' = the same kernel sources run through the host simulator
(CPU tensors, no GPU needed); 'gpu' = the hipcc-built gfx950 library on cuda:0 (pytest -m gpu)."""
import pytest
import torch


class Backend:
    def __init__(self, name):
        self.name = name
        if name == "emu":
            from hipemu.loader import load_emulator
            self.lib = load_emulator()
            self.device = torch.device("cpu")
        else:
            from mds.cabi import load
            self.lib = load()
            self.device = torch.device("cuda:0")

    def stream(self):
        if self.name == "emu":
            return 0
        return torch.cuda.current_stream().cuda_stream

    def call(self, op, args):
        self.lib.call(op, args, self.stream())

    def sync(self):
        if self.name != "emu":
            torch.cuda.synchronize()

    def t(self, x, dtype=None):
        x = x.to(self.device)
        if dtype is not None:
            x = x.to(dtype)
        return x.contiguous()


BACKENDS = ["emu", pytest.param("gpu", marks=pytest.mark.gpu)]
_cache = {}


@pytest.fixture(params=BACKENDS)
def be(request):
    name = request.param
    if name not in _cache:
        _cache[name] = Backend(name)
    return _cache[name]


@pytest.fixture
def be_gpu():
    """the gfx950 library only: for tests that never run on the simulator (it is then not even built on the GPU box)"""
    if "gpu" not in _cache:
        _cache["gpu"] = Backend("gpu")
    return _cache["gpu"]


DT = {"f32": (0, torch.float32), "bf16": (1, torch.bfloat16)}


def tol(dt):
    return dict(rtol=2e-4, atol=2e-5) if dt == "f32" else dict(rtol=2e-2, atol=2e-2)


def assert_close(got, want, dt, scale=1.0, msg=""):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    t = tol(dt)
    err = (got - want).abs()
    bound = t["atol"] * scale * max(1.0, want.abs().max().item()) + t["rtol"] * want.abs()
    bad = err > bound
    assert not bad.any(), (f"{msg}: {int(bad.sum())}/{bad.numel()} mismatches, max err {err.max().item():.3e} "
                           f"(ref max {want.abs().max().item():.3e}) first bad idx {bad.nonzero()[0].tolist()}")
