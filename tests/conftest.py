import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped automatically when no device is visible, so that a
    # plain `pytest tests/` on the CPU container stays green.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library is a build artefact (git-ignored): build it once if this checkout has none and a hipcc is
    around (the CPU container cross-compiles gfx950; on the GPU box the prebuilt file travels with the snapshot)."""
    from openvivqa_amd import build as B
    if not os.path.exists(B.LIB):
        try:
            B.build(verbose=False)
        except Exception as exc:  # noqa: BLE001 -- the tests that need it will say so themselves
            print(f"conftest: could not build {B.LIB}: {exc}", file=sys.stderr)
    yield


# files whose tests run with guard bands around every buffer the C-ABI wrappers allocate (tests/redzone.py); the multi-step
# training tests are left out (every eager step's allocations would stay alive until the end of the test)
_RED_ZONE_FILES = ("test_kernels_gpu.py", "test_blocks_gpu.py", "test_modules_gpu.py")


@pytest.fixture(autouse=True)
def _red_zones(request, monkeypatch):
    """4-KiB bands of 0xFF around the outputs / workspaces ``openvivqa_amd.ops`` allocates and around the inputs a test
    module's ``rnd()`` makes: an out-of-bounds store of any kernel fails the test that ran it (GPU AddressSanitizer is
    not available on this pool).  OVQA_TEST_REDZONE=0 switches the bands off."""
    if (os.path.basename(str(request.node.fspath)) not in _RED_ZONE_FILES or "gpu" not in request.node.keywords
            or os.environ.get("OVQA_TEST_REDZONE", "1") == "0"):
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    import redzone
    rz = redzone.install(monkeypatch, request.module)
    yield
    torch.cuda.synchronize()
    rz.check()


def parity_record(test, what, value, bar):
    """Measured parity figures: with OVQA_PARITY_REPORT=<file> every compared quantity is appended as a TSV line
    (test, quantity, measured error, bar), so that the bars written in the tests can be checked against the spread
    actually measured on the GPU (profiles/README.md quotes the file)."""
    path = os.environ.get("OVQA_PARITY_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(f"{test}\t{what}\t{value:.4e}\t{bar:.1e}\n")
    return value


# no gradient tensor passes further than this from the emulation, whatever the format's own error.  Round 5: 0.15 -> 0.12
# (measured worst at the depth of the deepest shipped stack, 30 chained blocks at fresh weights: 0.110, profiles/
# r05_parity_report.tsv); the one case that needed 0.15 -- CoAttentionEncoder at L = 6, 48 chained blocks, no shipped
# config -- no longer goes through the clause at all (tests/test_modules_gpu.py: tensors the bf16 format itself cannot
# resolve there are counted, not asserted)
ESCAPE_CEILING = 0.12
ESCAPES = []           # (test tag, tensor, measured, bar, format error) of every pass through the second arm


def grad_close(e_hip_vs_emu, e_emu_vs_fp32, bar, tag=None, what=None, allow_escape=True):
    """Criterion for ONE gradient tensor of the bf16 mode.  e_hip_vs_emu: relative L2 distance of the HIP gradient from
    the bf16-emulating oracle's; e_emu_vs_fp32: distance of the emulating oracle's from the fp32 oracle's -- the error
    the bf16 storage format itself puts on this tensor, no kernel involved.  A tensor passes when the HIP path is within
    ``bar`` of the emulation, or -- for gradients that are cancellations far below their terms (fc_q / fc_k behind a
    near-uniform softmax: up to 3000x smaller than the FFN gradients of the same layer), which the format itself cannot
    resolve to ``bar`` -- within TWICE the format's own error AND within ESCAPE_CEILING (round 4: the second arm had no
    ceiling).  Every pass through the second arm is listed in ESCAPES and, with OVQA_PARITY_REPORT, written to the report
    as an ``[escape clause]`` row; ``allow_escape=False`` (the non-degenerate operating points) disables it."""
    if e_hip_vs_emu < bar:
        return True
    if allow_escape and e_hip_vs_emu < 2.0 * e_emu_vs_fp32 and e_hip_vs_emu < ESCAPE_CEILING:
        ESCAPES.append((tag, what, e_hip_vs_emu, bar, e_emu_vs_fp32))
        if tag is not None:
            parity_record(tag, f"[escape clause] {what} (format's own error {e_emu_vs_fp32:.3e})", e_hip_vs_emu, bar)
        return True
    return False
