"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/uvaia_gpu.h declares, and refuses to
run without a GPU instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

from uvaia_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    capi.build_library()
    return capi.load_library()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "uvaia_gpu.h")).read()
    declared = set(re.findall(r"\b(uvaia_gpu_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_align_header_symbols_all_exported(lib):
    from uvaia_amd import align
    hdr = open(os.path.join(ROOT, "include", "uvaia_align.h")).read()
    declared = set(re.findall(r"\b(uvaia_align_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(align.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_aligner_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from uvaia_amd import align
    with pytest.raises(align.AlignError) as ei:
        align.Aligner(b"ACGTACGTAC")
    assert ei.value.code == -2          # UVAIA_ALIGN_ENODEV


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "uvaia_gpu.h"\n#include "uvaia_align.h"\nint main(void){ uvaia_gpu_ctx *c = 0; uvaia_aligner *a = 0; (void)c; (void)a; return UVAIA_GPU_OK + UVAIA_ALIGN_OK; }\n')
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    seqs = [b"ACGTACGTAC"]
    with pytest.raises(capi.GpuError) as ei:
        capi.Engine(seqs, b"ACGTACGTAC", list(range(10)), [], [], nbest=2, max_pool=4)
    assert ei.value.code == -2          # UVAIA_GPU_ENODEV: there is no CPU fallback


def test_product_never_touches_the_oracle():
    """The product tree must not reference oracle/ (tests, smoke and bench's cpu_baseline leg are the only users)."""
    bad = []
    for base in ("uvaia_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if os.sep + "build" in dp or os.sep + "lib" in dp or "__pycache__" in dp:
                continue
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle_lib|liboracle|orc_[a-z]+\s*\(|#include\s+\"[^\"]*oracle", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
