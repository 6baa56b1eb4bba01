"""Host-side AddressSanitizer build of the library (SURVEY §5 "optional -fsanitize=address host build"):
`make -C jchemo.jl_amd/csrc asan` -> lib/libjchemo_hip_asan.so (device code NOT instrumented: GPU ASan is not available on
this pool).  Skipped unless that library has been built.  Without a GPU the reachable host code is the entry-point
plumbing: version, context creation failing cleanly, argument validation, error strings."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "jchemo.jl_amd", "lib", "libjchemo_hip_asan.so")

SCRIPT = r"""
import ctypes as C, sys
lib = C.CDLL(sys.argv[1])
lib.jch_version.restype = C.c_int32
assert lib.jch_version() >= 101
lib.jch_last_error.restype = C.c_char_p
h = C.c_void_p()
st = lib.jch_ctx_create(C.byref(h), C.c_int32(0), None, C.c_uint32(0))
if st != 0:                                   # no GPU here: a clean failure with a message, no context
    assert not h.value
else:                                         # (a GPU box) exercise validation on a live context, then destroy it
    assert lib.jch_ctx_get_counter(h, C.c_int32(99), None) != 0
    assert len(lib.jch_last_error(h)) > 0
    assert lib.jch_ctx_destroy(h) == 0
# NULL-context calls must be rejected, not dereferenced
assert lib.jch_plskern_fit(None, None, None, C.c_int64(0), None, C.c_int64(0), None, None, None, None, None, None, None, None, None, None, None, None, None) != 0
assert lib.jch_transform(None, C.c_int32(0), None, C.c_int64(0), C.c_int64(0), C.c_int64(0), None, None, None, C.c_int32(0), None, C.c_int64(0)) != 0
assert lib.jch_ctx_destroy(None) == 0
uid = C.create_string_buffer(128)
lib.jch_comm_unique_id(uid)                   # (dlopens RCCL if present; any status is fine, it must not corrupt memory)
print("asan-ok")
"""


@pytest.mark.skipif(not os.path.exists(LIB), reason="make -C jchemo.jl_amd/csrc asan has not been run")
def test_host_entry_points_under_address_sanitizer():
    rt = subprocess.run(["/opt/rocm/bin/hipcc", "--print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:verify_asan_link_order=0")
    r = subprocess.run([sys.executable, "-c", SCRIPT, LIB], capture_output=True, text=True, env=env, timeout=300)
    assert "ERROR: AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "asan-ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
