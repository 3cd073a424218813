"""The C-ABI library loads and exports exactly what include/mi355rec.h declares (CPU only:
no compute call is made)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'mi355rec.h')


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mi_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from recommendation_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail("libmi355rec.so is not built: run __graft_entry__.build() (make -C recommendation_amd/csrc)")
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "header declares %s but the library does not export it" % n
    assert sorted(_lib.EXPORTS) == names, "ctypes signature table and header are out of step"
    assert lib.mi_version().decode().startswith('mi355rec')


def test_every_entry_point_cites_the_reference_or_says_it_has_no_counterpart():
    text = open(HEADER).read()
    assert 'r_model.py' in text and 'model_trainer.py' in text


def test_host_side_argument_checks_need_no_gpu():
    """Entry points validate shapes on the host before touching the device."""
    from recommendation_amd import _lib
    lib = _lib.load()
    rc = lib.mi_embag_fwd(None, None, 1, 4, None, 16, None, 1, 8, 0, 0, None, None, 0, 4, None, None)
    assert rc == -1 and b'null pointer' in lib.mi_last_error()
    rc = lib.mi_sort_pairs_u32(None, None, 5, 40, None, 0, None)
    assert rc == -1 and b'key_bits' in lib.mi_last_error()
    assert lib.mi_embag_bwd_workspace_bytes(0) > 0
    assert lib.mi_embag_bwd_workspace_bytes(1 << 20) >= 2 * 4 * (1 << 20)


@pytest.mark.gpu
def test_binding_loaded_before_torch_still_reaches_the_device():
    """A process that loads the binding FIRST and imports torch afterwards (build() followed by smoke() in one
    interpreter) must still launch on the GPU: torch ships its own HIP runtime, and two runtimes in one process
    leave this library's launches with "no ROCm-capable device"."""
    import subprocess
    import sys
    code = ("from recommendation_amd import _lib; _lib.load(); import torch; from recommendation_amd import ops; "
            "d = torch.device('cuda:0'); x = torch.arange(1024, dtype=torch.float32, device=d); "
            "out = torch.empty((1,), dtype=torch.float32, device=d); ops.reduce_sum(x, 1.0, out=out); "
            "assert float(out.item()) == 1023 * 1024 / 2; print('ok')")
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stderr[-2000:]
