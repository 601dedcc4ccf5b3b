"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/*.h declares.
No compute call is made here (there is no GPU in this tier)."""
import ctypes, os, re
import numpy as np
import aleo_amd
from aleo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for fn in os.listdir(os.path.join(ROOT, 'include')):
        if not fn.endswith('.h'): continue
        src = open(os.path.join(ROOT, 'include', fn)).read()
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        names += re.findall(r'\b(aleo_mi355x_[a-z0-9_]+)\s*\(', src)
    return sorted(set(names))


def test_header_and_library_agree():
    L = aleo_amd.lib()
    decl = declared_functions()
    assert len(decl) >= 15
    for name in decl:
        assert hasattr(L, name), f'{name} declared in include/ but not exported by libaleo_mi355x.so'
    assert sorted(_lib.EXPORTS) == decl, 'python binding list and header drifted apart'


def test_strerror_and_version_do_not_need_a_gpu():
    L = aleo_amd.lib()
    assert L.aleo_mi355x_version().startswith(b'aleo_mi355x')
    assert L.aleo_mi355x_strerror(0) == b'ok'
    assert L.aleo_mi355x_strerror(2) == b'bad argument'
    assert b'unknown' in L.aleo_mi355x_strerror(12345)


def test_missing_library_fails_loudly(monkeypatch):
    import pytest
    monkeypatch.setattr(_lib, '_LIB', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libaleo_mi355x.so')
    with pytest.raises(_lib.AleoMi355xError):
        _lib.lib()


def test_g1_sum_host_tail_matches_oracle(oracle):
    """aleo_mi355x_g1_sum is the host-side group add that follows the all-gather of per-GPU partials: pure CPU."""
    from oracle import pyref as p
    c = oracle
    G1 = c.affine_from_ints([p.G1_GENERATOR])[0]
    B = c.g1_multiples(G1, 8)
    parts = []
    for i in range(8):   # Jacobian (x, y, 1) of (i+1)G
        one = c.fq_to_mont(c.ints_to_limbs([1], 6)).reshape(6)
        parts.append(np.concatenate([np.ascontiguousarray(B[i, :96]).view(np.uint64), one]))
    parts.append(np.concatenate([one, one, np.zeros(6, dtype=np.uint64)]))     # identity (1,1,0)
    got = aleo_amd.g1_sum(np.stack(parts))
    assert c.jac_to_int_point(got) == p.g1_mul(p.G1_GENERATOR, 36)
    same = aleo_amd.g1_sum(np.stack([parts[2], parts[2]]))                      # doubling branch
    assert c.jac_to_int_point(same) == p.g1_mul(p.G1_GENERATOR, 6)
    neg = c.affine_from_ints([p.g1_neg(p.g1_mul(p.G1_GENERATOR, 3))])[0]
    pneg = np.concatenate([np.ascontiguousarray(neg[:96]).view(np.uint64), one])
    assert c.jac_to_int_point(aleo_amd.g1_sum(np.stack([parts[2], pneg]))) is None
    assert c.jac_to_int_point(aleo_amd.g1_sum(np.zeros((0, 18), dtype=np.uint64))) is None


def test_evaluation_domain_host_logic():
    d = aleo_amd.EvaluationDomain(5)
    assert d.size == 8 and d.log_size_of_group == 3
    assert aleo_amd.EvaluationDomain(1).size == 1
    assert aleo_amd.EvaluationDomain.new(1 << 48) is None          # beyond the two-adicity of Fr: reference returns None


def build_cpp_host_mirror(tmpdir, name='host_mirror_test'):
    """g++ a C++ caller of the C ABI (tests/cpp/<name>.cpp) against the in-tree library; returns the binary path."""
    import subprocess
    exe = os.path.join(str(tmpdir), name)
    libdir = os.path.join(ROOT, 'aleo_amd', 'lib')
    subprocess.check_call(['g++', '-std=c++17', '-O2', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'tests', 'cpp', name + '.cpp'), '-o', exe,
                           '-L', libdir, '-laleo_mi355x', '-Wl,-rpath,' + libdir])
    return exe


def test_cpp_host_mirror_compiles_and_links(tmp_path):
    """include/aleo_mi355x.hpp (the C++ mirror of VariableBase / EvaluationDomain) builds and links against the C ABI;
    the program itself needs a GPU and is run by tests/test_gpu_parity.py::test_cpp_host_mirror."""
    exe = build_cpp_host_mirror(tmp_path)
    assert os.path.exists(exe)
    assert os.path.exists(build_cpp_host_mirror(tmp_path, 'varuna_prove_test'))      # the whole-proof entry points from plain C++


def test_routing_thresholds_and_their_environment_overrides():
    """aleo_mi355x_min_msm / _min_ntt (the sizes from which the Rust arms route to the GPU): the measured defaults, and
    ALEO_MI355X_MIN_MSM / ALEO_MI355X_MIN_NTT read per call (no GPU involved)."""
    import os, subprocess, sys
    code = ("import aleo_amd; L = aleo_amd.lib(); print(int(L.aleo_mi355x_min_msm()), int(L.aleo_mi355x_min_ntt()))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def run(extra):
        env = dict(os.environ, PYTHONPATH=root, **extra); env.pop('ALEO_MI355X_MIN_MSM', None) if 'ALEO_MI355X_MIN_MSM' not in extra else None
        if 'ALEO_MI355X_MIN_NTT' not in extra: env.pop('ALEO_MI355X_MIN_NTT', None)
        return subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300).stdout.split()
    assert run({}) == ['1024', '4096']
    assert run({'ALEO_MI355X_MIN_MSM': '65536', 'ALEO_MI355X_MIN_NTT': '32'}) == ['65536', '32']
    assert run({'ALEO_MI355X_MIN_MSM': 'nonsense'}) == ['1024', '4096']


# ---- INTEGRATION.md's Rust stub against the header (row f2: the only form the Rust side can take in this image) ---------------------------------------------
_C2RUST = {'int32_t': 'i32', 'uint32_t': 'u32', 'uint64_t': 'u64', 'size_t': 'usize', 'int64_t': 'i64', 'double': 'f64', 'uint8_t': 'u8'}


def _c_prototypes():
    """name -> (return type, [normalised parameter types]) for every function include/aleo_mi355x.h declares."""
    src = open(os.path.join(ROOT, 'include', 'aleo_mi355x.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(int32_t|size_t|const char\*)\s+(aleo_mi355x_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, params = m.group(1), m.group(2), ' '.join(m.group(3).split())
        ps = [] if params in ('', 'void') else [p.strip() for p in params.split(',')]
        out[name] = (ret, ps)
    return out


def _c_param_kind(p):
    """'ptr' for any pointer / array parameter, else the Rust scalar name of the C integer type."""
    if '*' in p or '[' in p: return 'ptr'
    t = re.sub(r'\bconst\b', '', p).split()
    return _C2RUST.get(t[0], t[0])


def _rust_param_kind(p):
    t = p.split(':', 1)[1]
    t = re.sub(r'/\*.*?\*/', '', t).strip()
    return 'ptr' if t.startswith('*') else t


def test_integration_md_rust_stub_matches_the_header():
    """Every `fn aleo_mi355x_*` of the extern "C" blocks in INTEGRATION.md must name a function the header declares, with the same number of parameters,
    a pointer where the header has a pointer, the same integer width elsewhere, and an i32 status where the header returns int32_t."""
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    protos = _c_prototypes()
    decls = re.findall(r'\bfn\s+(aleo_mi355x_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([A-Za-z0-9_*\s]+?))?\s*;', md, flags=re.S)
    assert len(decls) >= 19
    for name, params, ret in decls:
        assert name in protos, f'INTEGRATION.md binds {name}, which include/aleo_mi355x.h does not declare'
        c_ret, c_params = protos[name]
        params = re.sub(r'//[^\n]*', '', params)
        rp = [p.strip() for p in re.sub(r'/\*.*?\*/', '', params, flags=re.S).split(',') if p.strip()]
        assert len(rp) == len(c_params), (name, rp, c_params)
        for r, c_ in zip(rp, c_params):
            assert _rust_param_kind(r) == _c_param_kind(c_), (name, r, c_)
        want_ret = {'int32_t': 'i32', 'size_t': 'usize', 'const char*': None}[c_ret]
        if want_ret: assert ret.strip() == want_ret, (name, ret, c_ret)
