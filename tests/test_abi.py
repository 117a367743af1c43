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


def test_tuning_and_query_structs_match_the_header(tmp_path):
    """ctypes mirrors of uvaia_gpu_tuning and uvaia_gpu_query: same field names in the same order as include/uvaia_gpu.h, same sizes and
    offsets as a C compiler gives them (a field added on one side only would silently shift every knob after it)."""
    import subprocess
    hdr = open(os.path.join(ROOT, "include", "uvaia_gpu.h")).read()

    def fields_of(struct_name):
        end = re.search(r"\}\s*%s\s*;" % struct_name, hdr).start()
        body = hdr[hdr.rindex("typedef struct", 0, end):end].split("{", 1)[1]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            for part in decl.split(","):                       # "const size_t *idx_c, *idx_m, *idx": one name per part, the last identifier
                ids = re.findall(r"[A-Za-z_][A-Za-z_0-9]*", re.sub(r"\[.*?\]", "", part))
                if ids:
                    names.append(ids[-1])
        return names

    for cname, ctype in (("uvaia_gpu_tuning", capi.Tuning), ("uvaia_gpu_query", capi._Query)):
        names = fields_of(cname)
        assert names == [f[0] for f in ctype._fields_], cname
        src = tmp_path / ("%s.c" % cname)
        src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "uvaia_gpu.h"\nint main(void){ printf("%zu", sizeof(' + cname + '));'
                       + "".join(' printf(" %%zu", offsetof(%s, %s));' % (cname, n) for n in names) + " return 0; }\n")
        exe = tmp_path / cname
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
        got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
        assert got[0] == C.sizeof(ctype), cname
        assert got[1:] == [getattr(ctype, n).offset for n in names], cname
