"""CPU: the host C++ of the batched pipeline under ThreadSanitizer and AddressSanitizer + UBSan (hostpool.cpp's chunk claiming shared by
the bench's worker-group threads, lsap.cpp, pyset.cpp, host_phases.h).  GPU sanitizers are not available on the pool; the device code
has no host-visible shared state beyond what these files hold."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'deepdish_amd', 'csrc')
SRCS = [os.path.join(ROOT, 'tests', 'sanitize', 'host_harness.cpp')] + [os.path.join(CSRC, f) for f in ('hostpool.cpp', 'lsap.cpp', 'pyset.cpp')]


@pytest.mark.parametrize('name,flags,steps', [('tsan', ['-fsanitize=thread'], 120), ('asan_ubsan', ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined'], 300)])
def test_host_code_is_clean_under_sanitizers(tmp_path, name, flags, steps):
    exe = str(tmp_path / ('harness_' + name))
    cmd = ['g++', '-std=c++17', '-O1', '-g', '-fno-omit-frame-pointer', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I' + os.path.join(ROOT, 'include')] + flags + SRCS + ['-lpthread', '-o', exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, DD_HOST_THREADS='6', TSAN_OPTIONS='halt_on_error=1 exitcode=66', ASAN_OPTIONS='detect_leaks=0')
    r = subprocess.run([exe, str(steps)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and '0 mismatches' in r.stdout and 'WARNING: ThreadSanitizer' not in r.stderr and 'ERROR: AddressSanitizer' not in r.stderr \
        and 'runtime error' not in r.stderr, (r.stdout[-500:], r.stderr[-3000:])
