"""World-size-2 gloo test of the sharded-MSM path (CPU): shard ranges, the 144-byte all-gather and the local
group add.  The per-rank partial comes from the oracle here (no GPU in this tier); on the GPU box the same
aleo_amd.dist functions are fed by the HIP MSM (tests/test_gpu_parity.py::test_sharded_msm_two_shards)."""
import os, sys, socket
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close(); return port


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    import aleo_amd
    from aleo_amd import dist as adist
    from oracle import coracle as c
    import util
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        B = util.multiples_bases(n); S = util.uniform_scalars(n, 4242)
        lo, hi = adist.shard_range(n, rank, world)
        total = adist.sharded_msm(lambda: c.msm_g1(B[lo:hi], S[lo:hi], threads=1, variant=1), aleo_amd.g1_sum)
        q.put((rank, lo, hi, total.tolist()))
    finally:
        dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    from aleo_amd.dist import shard_range
    for n in (0, 1, 7, 8, 1000, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]; assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_sharded_msm_world2_gloo():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import coracle as c
    import util
    n, world, port = 301, 2, _free_port()
    ctx = mp.get_context('spawn'); q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p_ in procs: p_.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p_ in procs: p_.join(60); assert p_.exitcode == 0
    res.sort()
    assert res[0][3] == res[1][3], 'ranks disagree on the combined result'
    got = c.jac_to_int_point(np.array(res[0][3], dtype=np.uint64))
    assert got == util.expected_multiples_msm(util.uniform_scalars(n, 4242), n)
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 151, 151, 301)


def _ntt_worker(rank, world, port, lg_n, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    from aleo_amd import dist as adist
    from oracle import coracle as c
    import util
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        x = c.fr_to_mont(util.uniform_scalars(1 << lg_n, 777))
        dom = adist.ShardedDomain(lg_n, rank, world, ops=util.OracleLocalOps(), lg_rows=lg_n // 2 + 1)     # R != C on purpose
        out = {}
        for coset in (False, True):
            mine = torch.from_numpy(dom.coefficient_shard(x).view(np.int64).copy())
            ev = dom.forward(mine.clone(), coset=coset)
            back = dom.inverse(ev.clone(), coset=coset)
            out[coset] = (ev.numpy().view(np.uint64).tolist(), bool((back == mine).all()))
        q.put((rank, dom.evaluation_indices().tolist(), out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_ntt_world2_gloo():
    """4-step NTT over two ranks (one all-to-all each way): every rank's evaluation block equals the oracle's
    single-domain transform at the natural indices it claims to hold, for fft and coset_fft, and the inverse returns the
    rank's coefficient block bit for bit."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import coracle as c
    import util
    lg_n, world, port = 7, 2, _free_port()
    ctx = mp.get_context('spawn'); q = ctx.Queue()
    procs = [ctx.Process(target=_ntt_worker, args=(r, world, port, lg_n, q)) for r in range(world)]
    for p_ in procs: p_.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p_ in procs: p_.join(60); assert p_.exitcode == 0
    x = c.fr_to_mont(util.uniform_scalars(1 << lg_n, 777))
    seen = set()
    for rank, idx, out in res:
        idx = np.array(idx)
        for coset in (False, True):
            full = c.ntt_fr(x, 0, 0, 1 if coset else 0)
            ev, round_trip_ok = out[coset]
            assert round_trip_ok, (rank, coset)
            assert (np.array(ev, dtype=np.uint64) == full[idx]).all(), (rank, coset)
        seen |= set(idx.reshape(-1).tolist())
    assert seen == set(range(1 << lg_n))          # the evaluation layout covers the domain exactly once


@pytest.mark.timeout(300)
def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2 ...` with no torchrun and no WORLD_SIZE in the environment (how the driver starts the N = 1 bench): the parent must
    spawn the two ranks itself, stay off the GPU, relay rank 0's ONE JSON line with n_gpus = 2, and exit 0.  --launch-check keeps it arithmetic-free,
    so it runs here; the same launcher with the real MSM step is a GPU test (tests/test_gpu_parity.py)."""
    import json, subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--launch-check', '--steps', '2'],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks'] == 2 and out['gather_in_rank_order'] and out['value'] is None
    assert out['config4_per_rank_gather'] is True          # the per-rank exchange of the 2^26-point MSM the default N > 1 run also reports (config4_split)


@pytest.mark.timeout(300)
def test_bench_launcher_reports_a_failing_rank():
    """A rank that dies takes the launch down with its status instead of leaving the others at a barrier: without a GPU the real (non --launch-check)
    step refuses to run in every rank, and the launcher must come back non-zero and print no JSON line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    env['HIP_VISIBLE_DEVICES'] = ''; env['ROCR_VISIBLE_DEVICES'] = ''
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]
