"""The drop-in boundary from compiled code, not ctypes: a C99 program (tests/c/boundary_roundtrip.c) and the
two C++ classes INTEGRATION.md shows, extracted from the markdown so the snippets cannot rot, are compiled
against include/xsi_hip.h and linked with libxsi_hip.so (CPU part), then run on the GPU (-m gpu part)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "xsqueezeit_amd")


def _build_c(tmp):
    exe = os.path.join(tmp, "boundary_roundtrip")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-Werror",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "boundary_roundtrip.c"),
                           "-o", exe, "-L", LIBDIR, "-lxsi_hip", "-Wl,-rpath," + LIBDIR])
    return exe


def _extract_integration_snippets(tmp):
    """```cpp blocks of INTEGRATION.md whose first line names a header file -> files of that name."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    found = {}
    for block in re.findall(r"```cpp\n(.*?)```", md, flags=re.S):
        m = re.match(r"// (\w+\.hpp) ", block)
        if m:
            found[m.group(1)] = block
            with open(os.path.join(tmp, m.group(1)), "w") as f:
                f.write("#pragma once\n" + block)
    assert set(found) == {"accessor_internals_hip.hpp", "xsi_factory_hip.hpp"}, sorted(found)
    return found


def _build_cxx(tmp):
    _extract_integration_snippets(tmp)
    exe = os.path.join(tmp, "integration_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror",
                           "-I", tmp, "-I", os.path.join(ROOT, "tests", "cxx", "ref_decls"),
                           "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", "integration_main.cpp"),
                           "-o", exe, "-L", LIBDIR, "-lxsi_hip", "-Wl,-rpath," + LIBDIR])
    return exe


def test_c_program_and_integration_classes_compile_and_link(tmp_path):
    assert os.path.exists(_build_c(str(tmp_path)))
    assert os.path.exists(_build_cxx(str(tmp_path)))


@pytest.mark.gpu
def test_c_program_round_trip_on_gpu(tmp_path):
    exe = _build_c(str(tmp_path))
    out = subprocess.run([exe, str(tmp_path / "c.xsi"), "2504", "20000", "8192"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok lines=20000 haps=5008 bad_lines=0"), out.stdout
    # multi-batch, ragged: 3 blocks per writer batch, last batch short
    env = dict(os.environ, XSI_WRITER_BATCH_BLOCKS="3")
    out = subprocess.run([exe, str(tmp_path / "c2.xsi"), "150", "5000", "512"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr


@pytest.mark.gpu
def test_integration_classes_round_trip_on_gpu(tmp_path):
    exe = _build_cxx(str(tmp_path))
    out = subprocess.run([exe, str(tmp_path / "i.xsi")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=600)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
