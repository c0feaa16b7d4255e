"""Build libdeepdish_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m deepdish_amd.build [--force]

Objects are cached under deepdish_amd/csrc/_obj and rebuilt when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libdeepdish_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'

COMMON = ['--offload-arch=' + ARCH, '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
          '-I', os.path.join(HERE, '..', 'include')]
# The deep_sort math is parity-checked to the last bits in f64: keep a*b+c un-fused there.
STRICT_FP = {'kalman.hip', 'cost.hip', 'nms.hip', 'tracker.hip', 'lsap.cpp', 'pyset.cpp', 'api.hip', 'image.hip', 'pipeline.hip', 'mog2.hip', 'post.hip'}


# Per-file code generation flags.  netsq.hip: MFMA results in ordinary VGPRs -- its kernels are bound by vector-instruction issue and every
# accumulator parked in an AGPR costs a v_accvgpr_read before the requantisation (3144 of them in the file's kernels without the flag).
EXTRA = ({'netsq.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form'], 'netsq_front.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form'], 'netsq_mid.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form'], 'mars_tail.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form'],
          'mars_pair.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}
         if os.environ.get('DD_NO_VGPR_FORM') != '1' else {})


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(HERE, '..', 'include', 'deepdish_hip.h'))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force, hmt):
    obj = os.path.join(OBJ, src + '.o')
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(path), hmt, os.path.getmtime(__file__))):
        return obj, False
    cmd = [HIPCC] + COMMON + (['-ffp-contract=off'] if src in STRICT_FP else []) + EXTRA.get(src, []) + ['-x', 'hip', '-c', path, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hmt = _headers_mtime()
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, hmt), srcs))
    objs = [o for o, _ in res]
    rebuilt = any(c for _, c in res)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    if verbose:
        print('libdeepdish_hip.so: %s (%d sources, %s)' % (LIB, len(srcs), 'rebuilt' if rebuilt else 'up to date'))
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
