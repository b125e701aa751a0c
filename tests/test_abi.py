"""The C-ABI library loads (no GPU needed) and exports every symbol that
include/sykepic_hip.h declares; the ctypes table covers the same set."""

import ctypes
import re
from pathlib import Path

import pytest

from sykepic_hip import lib

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "sykepic_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spk_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert len(names) >= 20
    assert lib.LIB_PATH.is_file(), "run __graft_entry__.build() first"
    so = ctypes.CDLL(str(lib.LIB_PATH))
    for n in names:
        assert hasattr(so, n), f"{n} declared in the header but not exported"
    assert sorted(lib.SYMBOLS) == names
    lib.load()


def test_error_reporting_without_gpu_or_bad_args():
    so = lib.load()
    assert b"gfx950" in so.spk_version()
    h = ctypes.c_void_p()
    rc = so.spk_model_create(None, 0, 3, 50, 0, ctypes.byref(h))
    assert rc != 0 and so.spk_last_error()


def test_struct_layouts_match_header():
    assert ctypes.sizeof(lib.LayerDesc) == 11 * 4 + 4 + 96 + 96
    assert ctypes.sizeof(lib.OptimDesc) == 4 + 12 + 12 + 4 + 4 + 4 + 16
    assert ctypes.sizeof(lib.LayerTime) == 96 + 4 + 4 + 8 + 8  # 4 B padding before the doubles


def test_host_code_under_address_and_ub_sanitizers():
    """`csrc/build_asan.sh`: every translation unit with -fsanitize=address,undefined on its host pass + csrc/
    asan_driver.hip, which walks handle creation and its error paths, the parameter table (with a deliberately short
    key buffer), spk_last_error and the three tuner-cache parsers (valid, truncated, garbage and over-long lines).
    Without a GPU every HIP call fails and the error paths run; a sanitizer report aborts the driver (exit != 0)."""
    import os
    import subprocess
    csrc = ROOT / "syke-pic_amd" / "csrc"
    recipe = csrc / "build_asan.sh"
    if not recipe.is_file():
        pytest.skip("build_asan.sh is not shipped to GPU boxes (.gpurunignore): the sanitizer pass is a CPU-container test")
    exe = csrc / "build" / "asan" / "asan_driver"
    srcs = list(csrc.glob("*.hip")) + list(csrc.glob("*.h"))
    if not exe.is_file() or any(f.stat().st_mtime > exe.stat().st_mtime for f in srcs):
        subprocess.run(["bash", str(recipe)], check=True, timeout=1500)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("SPK_TUNE_CACHE", None)
    r = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "asan_driver: ok" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
