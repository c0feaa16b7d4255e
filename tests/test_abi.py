"""CPU: the C-ABI library loads without a GPU and exports every symbol include/deepdish_hip.h declares;
the product never imports the oracle."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'deepdish_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dd_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound():
    from deepdish_amd._lib import lib, MISSING, SIGNATURES, LIB_PATH
    l = lib()
    assert MISSING == []
    declared = _declared()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(l, name), 'declared in the header but not exported: ' + name
        assert name in SIGNATURES, 'declared in the header but not bound in _lib.py: ' + name
    out = subprocess.run(['nm', '-D', '--defined-only', LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (dd_[a-z0-9_]+)', out))
    assert set(declared) <= exported
    assert l.dd_version() >= 100


def test_errors_are_codes_not_exceptions():
    from deepdish_amd._lib import lib
    l = lib()
    assert l.dd_lsap_host(None, -1, 2, None, None) < 0
    assert b'dd_lsap_host' in l.dd_last_error()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'deepdish_amd')
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith('.py'):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(dp, fn)


def test_missing_library_is_a_loud_error(tmp_path):
    code = ("import deepdish_amd._lib as l; l.LIB_PATH = %r; l._lib = None\n"
            "try:\n    l.lib()\nexcept l.DeepDishHipError as e:\n    print('LOUD', e)\n" % str(tmp_path / 'nope.so'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=ROOT).stdout
    assert 'LOUD' in out and 'no CPU fallback' in out
