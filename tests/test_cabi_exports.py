"""CPU: the C-ABI library loads without a GPU and exports exactly what include/icap_hip.h declares."""
import ctypes
import os
import re

from image_captioning_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "icap_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(icap_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = declared_functions()
    assert len(names) >= 25
    assert sorted(_native.SIGNATURES) == names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_native.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in icap_hip.h but not exported"
    lib.icap_abi_version.restype = ctypes.c_int
    assert lib.icap_abi_version() == _native.ABI_VERSION
    assert _native.load().icap_abi_version() == _native.ABI_VERSION


def test_argument_checks_need_no_gpu():
    lib = _native.load()
    # null tensors are rejected before anything touches the device
    rc = lib.icap_conv2d_f32(None, None, None, None, None, None, None, 1, 1, 1, 4, 4, 1, 1, 1, 0, 0, 1, 1, 0, None)
    assert rc == 1 and b"null" in lib.icap_last_error()
    assert lib.icap_gemv_blocks(6756) > 0
    assert lib.icap_dwconv_chunks(112, 112, 32, 3, 1) > 0 and lib.icap_dwconv_chunks(7, 7, 1152, 5, 1) == 7
    assert lib.icap_expand_dwconv_chunks(56, 56, 16, 96, 3, 2) == 56 and lib.icap_expand_dwconv_chunks(14, 14, 112, 672, 5, 2) == -1
    # the stride-1 blocks at 14x14 / 7x7 run on the whole-image kernel: one chunk (the channel sums) per image
    assert lib.icap_expand_dwconv_chunks(14, 14, 80, 480, 3, 1) == 1 and lib.icap_expand_dwconv_whole_image(7, 7, 192, 1152, 5, 1) == 1
    assert lib.icap_expand_dwconv_whole_image(28, 28, 40, 240, 5, 1) == 0
    assert lib.icap_stem_dwconv_chunks(112, 112, 32) == 64
    assert lib.icap_greedy_tfm_ws_floats(8, 512, 2048, 6756) > 8 * 512 * 4


def test_library_was_built_against_this_header():
    """The Makefile bakes the POSIX cksum of include/icap_hip.h into the library; a library built from another version of
    the header (a changed signature under an unchanged name) is refused at load time."""
    from image_captioning_amd import _native as N
    import ctypes
    lib = ctypes.CDLL(N.LIB_PATH)
    lib.icap_header_hash.restype = ctypes.c_uint
    assert N.header_hash() is not None
    assert lib.icap_header_hash() == N.header_hash()
