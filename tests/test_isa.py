"""Compile-only checks on the device code (no GPU): the accumulation kernels' ISA must not store from a register that no instruction on any path
has written (tools/isa_undef_check.py) — the shape of the hipcc miscompile k_g2_accum28 met in round 4 (g2.hip: the `asm volatile` pin of the
accumulator), checked here for the G1 kernels with the same skip-the-loop structure as well, with and without the pin."""
import os, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import isa_undef_check as isa                                  # noqa: E402

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
CSRC = os.path.join(ROOT, 'aleo_amd', 'csrc')


def _asm(src, out, *defs):
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', '-I' + CSRC, *defs, os.path.join(CSRC, src), '-o', out],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def test_the_checker_sees_a_store_from_a_never_written_register():
    good = """
	.amdhsa_kernel k_demo
k_demo:
	v_mov_b32_e32 v3, 7
	s_cbranch_execz .LBB0_2
	v_mov_b32_e32 v4, v3
.LBB0_2:
	global_store_dword v[0:1], v3, off
	s_endpgm
.Lfunc_end0:
"""
    assert isa.undefined_reads(good) == {'k_demo': []}
    bad = good.replace('global_store_dword v[0:1], v3, off', 'global_store_dwordx2 v[0:1], v[4:5], off')      # v5 has no writer; v4 only on one path (fine for this check)
    rep = isa.undefined_reads(bad)['k_demo']
    assert len(rep) == 1 and rep[0][2] == ['v5']
    # a definition that only a loop body makes still counts as reaching the store behind the loop (union at the join): no false alarm
    loop = good.replace('s_cbranch_execz .LBB0_2\n\tv_mov_b32_e32 v4, v3\n.LBB0_2:', '.LBB0_1:\n\tv_mov_b32_e32 v9, v3\n\ts_cbranch_scc1 .LBB0_1\n').replace('v3, off', 'v9, off')
    assert isa.undefined_reads(loop) == {'k_demo': []}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='no hipcc')
def test_accumulation_kernels_store_only_written_registers():
    with tempfile.TemporaryDirectory() as d, ThreadPoolExecutor(3) as ex:
        jobs = {'msm': ex.submit(_asm, 'msm.hip', os.path.join(d, 'msm.s')),
                'g2': ex.submit(_asm, 'g2.hip', os.path.join(d, 'g2.s')),
                'g2_nopin': ex.submit(_asm, 'g2.hip', os.path.join(d, 'g2n.s'), '-DALEO_G2_NO_PIN')}
        want = {'msm': ('k_accum28', 'k_tree_pass', 'k_bucket_chunks', 'k_prog_pass', 'k_prog_final'), 'g2': ('k_g2_accum28', 'k_g2p_'), 'g2_nopin': ('k_g2_accum28',)}
        for key, fut in jobs.items():
            rep = isa.undefined_reads(fut.result(), want[key])
            assert rep, key                                        # the kernels were found
            bad = {k: v for k, v in rep.items() if v}
            assert not bad, (key, bad)
        assert any('k_accum28' in k for k in isa.undefined_reads(jobs['msm'].result(), ('k_accum28',)))
