#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X prover backend (BASELINE.json metric, configs[1]):
standalone 2^20-point BLS12-377 G1 Pippenger MSM = one `aleo_mi355x_msm_g1_pinned` call per step — bases resident in
HBM (the SRS of a proving key is pinned once), the 32 MB of canonical scalars in HOST memory and uploaded inside the timed
region, result on the host.  The same line carries the variants beside it: scalars already resident, no fixed-base
table, the table's build time and size.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W [--config 4]

A step = one MSM pass: every rank runs the full Pippenger over its own shard of (scalar, base) pairs, then the 144-byte
partials are all-gathered (RCCL) and added locally.  Default: weak scaling, 2^20 pairs per GPU.
`--config 4` names BASELINE configs[4] instead: a 2^26-point MSM split over the ranks (2^23 per GPU at N = 8; strong
scaling).  value = points processed by all ranks / max-over-ranks wall time.  Prints ONE JSON line on rank 0."""
from __future__ import annotations
import argparse, json, os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')          # as aleo_amd/__init__.py does (read by the HIP runtime at its initialisation, which torch may trigger before the package is imported)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=1, choices=[1, 4],
                    help='BASELINE configs index: 1 = 2^20 points per GPU (weak scaling); 4 = 2^26 points split over the ranks (strong scaling)')
    ap.add_argument('--lg-n', type=int, default=None, help='log2 of the points per GPU (default 20; with --config 4: log2 of the TOTAL, default 26)')
    ap.add_argument('--scalars', default='uniform', choices=['uniform', 'witness'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-crossover', action='store_true', help='skip the one-shot GPU vs CPU crossover table of the cpu_baseline leg')
    ap.add_argument('--cpu-sample-lg', type=int, default=20)
    ap.add_argument('--no-precompute', action='store_true', help='headline without the fixed-base window table')
    ap.add_argument('--no-variants', action='store_true', help='skip the resident-scalars / no-table variants')
    ap.add_argument('--no-kzg-chain', action='store_true', help='skip the secondary 2^22 iNTT -> commit measurement (config[2])')
    ap.add_argument('--proof-proxy-lg', type=int, default=20, help='log2 constraints of the Varuna operator-schedule replay (0 = skip)')
    ap.add_argument('--proof-proxy-cpu-lg', type=int, default=15, help='size of the same replay on the CPU oracle (cpu_baseline leg)')
    ap.add_argument('--varuna-lg', type=int, default=15, help='log2 constraints of the AHP prover measurement (row a6; 0 = skip)')
    ap.add_argument('--varuna-big-lg', type=int, default=20, help='a second, large circuit for the prover: single proof and 8 instances only (0 = skip)')
    ap.add_argument('--varuna-cpu-lg', type=int, default=10, help='size at which the CPU restatement of the prover is timed and the device proof verified (cpu_baseline leg)')
    ap.add_argument('--concurrent-callers', type=int, default=4, help='secondary: aggregate rate with this many caller threads (0/1 = skip)')
    ap.add_argument('--sharded-ntt-lg', type=int, default=24, help='size of the secondary sharded-NTT measurement at N > 1 (stderr)')
    ap.add_argument('--config4-lg', type=int, default=26, help='at N > 1 (default --config 1 run): log2 of the ONE MSM of BASELINE configs[4] that is also split over the ranks and reported as config4_2^K in the same line (0 = skip)')
    ap.add_argument('--aux-sharded-ntt', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--in-process', action='store_true',
                    help='with --config 4 and no launcher: ONE process drives --gpus N device contexts through aleo_mi355x_msm_g1_sharded (the visible devices are listed cyclically when there are fewer than N)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="'gloo' is only for rehearsing the N>1 path with several ranks sharing one GPU")
    ap.add_argument('--launch-check', action='store_true',
                    help='N > 1 plumbing only (runs without a GPU): spawn / rendezvous / the 144-byte gather / the max-over-ranks reduction under --backend gloo; prints a line with value null')
    args = ap.parse_args()
    if args.aux_sharded_ntt:
        raise SystemExit(aux_sharded_ntt(args))
    if args.in_process:
        raise SystemExit(in_process_sharded(args))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Plain `python bench.py --gpus N` (no torchrun): this process becomes the launcher.  It has not imported torch or touched HIP and never does
        # (no exec either): it starts the N ranks as children, relays rank 0's JSON line and exits with the first non-zero child status.
        raise SystemExit(spawn_ranks(args.gpus))
    if args.launch_check:
        raise SystemExit(launch_check(args))

    import torch
    import torch.distributed as dist
    import aleo_amd
    from aleo_amd import synth
    from aleo_amd import dist as adist

    rank = int(os.environ.get('RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    dev_index = local_rank % torch.cuda.device_count()          # one rank per GPU (ranks share a GPU only under --backend gloo)
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    gather_dev = dev if args.backend == 'nccl' else None       # RCCL gathers device tensors; gloo gathers host tensors
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl' and int(os.environ.get('LOCAL_WORLD_SIZE', world)) > torch.cuda.device_count():
            raise SystemExit('bench: %d local ranks but %d visible GPU(s): RCCL needs one GPU per rank (rehearse on one card with --backend gloo)' % (int(os.environ.get('LOCAL_WORLD_SIZE', world)), torch.cuda.device_count()))
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
    L = aleo_amd.lib()
    aleo_amd._lib.check(L.aleo_mi355x_init_device(dev_index), 'init')

    strong = args.config == 4
    if strong:
        total = 1 << (args.lg_n if args.lg_n is not None else 26)
        lo, hi = adist.shard_range(total, rank, world)          # rank r owns points [lo, hi) of the one big MSM
        n, first = hi - lo, lo + 1
        if n * 13 >= (1 << 32):
            raise SystemExit('bench --config 4: %d points per GPU exceed one launch (2^32 (bucket, point) pairs); use more ranks or --lg-n' % n)
    else:
        lg = args.lg_n if args.lg_n is not None else 20
        n = 1 << lg; total = n * world; first = rank * n + 1
    gen = synth.generator_affine104()
    pb = aleo_amd.PinnedBases.generate_multiples(gen, first, n)           # P_i = (first + i) * G, generated in HBM
    precompute_s = None
    if not args.no_precompute:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pb.precompute()           # setup, outside the timed region: the SRS of a proving key is fixed (bases_pin contract)
        precompute_s = time.perf_counter() - t0
    info = pb.info()
    mk = synth.uniform_scalars if args.scalars == 'uniform' else synth.witness_like_scalars
    scalars = mk(n, 0xA1E00002 + rank)                                     # HOST memory (pageable, as a Rust Vec would be)
    gather = adist.PartialGather(world, gather_dev) if world > 1 else None

    def step():
        part = aleo_amd.VariableBase.msm(pb, scalars)                      # aleo_mi355x_msm_g1_pinned: upload + MSM + result on host
        if world > 1:
            return aleo_amd.g1_sum(gather(part))
        return part

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    acc_kernel_ms, phases = [], []
    barrier(); t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        tm = aleo_amd.last_msm_timing(); acc_kernel_ms.append(tm['accum_kernel_ms']); phases.append(tm)      # accum_kernel_ms: mean over the call's k_accum28 launches (accum_launches)
    barrier(); t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev if gather_dev is not None else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())

    # correctness gate: the result must equal k*G with k = sum_i s_i * (first+i) over all ranks (an O(n) identity)
    k = synth.weighted_scalar_sum(scalars, first)
    if world > 1:
        ks = gather(synth.int_to_limbs(k, 4))                                # 4 limbs per rank, same collective
        k = sum(synth.limbs_to_int(row) for row in ks) % synth.FR_MODULUS
    if not result_is_multiple_of_generator(synth, res, k):       # big-integer double-and-add below: independent of the HIP path
        raise SystemExit('bench: MSM result does not equal k*G — refusing to report a number for a wrong result')

    # variants of the same MSM, outside the timed region (every rank runs them so the ranks stay in step; rank 0 reports)
    variants = {}
    if not args.no_variants:
        d_scalars = torch.from_numpy(scalars.view(np.int64)).to(dev); torch.cuda.synchronize()
        reps = max(3, min(args.steps, 10))

        def timed(fn):
            fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): r_ = fn()
            dt = (time.perf_counter() - t0) / reps
            if not (np.asarray(r_) == (res if world == 1 else part_ref)).all(): raise SystemExit('bench: a variant disagrees with the headline result')
            return dt
        part_ref = aleo_amd.VariableBase.msm(pb, scalars)
        dt = timed(lambda: aleo_amd.VariableBase.msm_device(pb, d_scalars.data_ptr(), n))
        variants['value_scalars_resident'] = n / dt; variants['ms_scalars_resident'] = dt * 1e3
        if not args.no_precompute and n <= (1 << 22):
            pb2 = aleo_amd.PinnedBases.generate_multiples(gen, first, n)       # the same points, no table: the one-shot path's schedule
            dt = timed(lambda: aleo_amd.VariableBase.msm(pb2, scalars))
            variants['value_no_table'] = n / dt; variants['ms_no_table'] = dt * 1e3
            pb2.close()
        if world == 1 and n <= (1 << 22):
            # the two-line drop-in of INTEGRATION.md 2 as it stands: plain aleo_mi355x_msm_g1 on host bases AND host scalars, nothing cached
            # (104 B/point uploaded and converted to the 28-bit rows inside every call, no table)
            host_bases = pb.download()
            dt = timed(lambda: aleo_amd.VariableBase.msm(host_bases, scalars))
            variants['value_one_shot_cold'] = n / dt; variants['ms_one_shot_cold'] = dt * 1e3
            del host_bases
        if world == 1 and not args.no_precompute and n <= (1 << 22):
            # witness-like scalars (60 % zero, 20 % one, 10 % < 2^16, 10 % uniform), resident: on the wide window, and on a 16-bit range
            # table over the whole set with the sparse hint (bases_precompute_range; built after every number above was taken)
            ws = synth.witness_like_scalars(n, 0xA1E00002); d_ws = torch.from_numpy(ws.view(np.int64)).to(dev); torch.cuda.synchronize()
            def timed_w(fn):
                r0 = fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(reps): r_ = fn()
                dt_ = (time.perf_counter() - t0) / reps
                if not (np.asarray(r_) == np.asarray(r0)).all(): raise SystemExit('bench: unstable witness-like result')
                return dt_, r0
            dt_wide, r_wide = timed_w(lambda: aleo_amd.VariableBase.msm_device(pb, d_ws.data_ptr(), n))
            t0 = time.perf_counter(); pb.precompute_range(0, n, 16); range_s = time.perf_counter() - t0
            dt_rng, r_rng = timed_w(lambda: aleo_amd.VariableBase.msm_device(pb, d_ws.data_ptr(), n, sparse=True))
            dt_host, r_host = timed_w(lambda: aleo_amd.VariableBase.msm(pb, ws))
            if not ((np.asarray(r_wide) == np.asarray(r_rng)).all() and (np.asarray(r_wide) == np.asarray(r_host)).all()): raise SystemExit('bench: the range table changes the result')
            variants['witness_like'] = {'ms_resident_wide_window': dt_wide * 1e3, 'ms_resident_range_table': dt_rng * 1e3, 'ms_host_scalars_range_table': dt_host * 1e3,
                                        'value_resident_range_table': n / dt_rng, 'range_table_build_s': range_s, 'range_table_window_bits': 16}
            del d_ws
        del d_scalars

    cfg4 = None
    if world > 1 and not strong and args.config4_lg:
        # BASELINE configs[4] in the driver's plain `--gpus N` line: ONE 2^26-point MSM split over the ranks (strong scaling), after the weak-scaling steps
        pb.close(); pb = None
        cfg4 = config4_split(aleo_amd, synth, adist, dist, torch, rank, world, gather_dev, barrier, args.config4_lg, args.backend, no_precompute=args.no_precompute)
        pb = aleo_amd.PinnedBases.generate_multiples(gen, first, n)
        if not args.no_precompute: pb.precompute()
    replicas = None
    if world > 1 and args.varuna_lg:
        replicas = prove_replicas(synth, dist, args.varuna_lg, world, gather_dev, barrier)      # every rank takes part in its collectives, whatever happens locally

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = total * args.steps / elapsed if strong else world * n * args.steps / elapsed
        ak = float(np.mean(acc_kernel_ms)) * 1e-3
        launches = int(round(float(np.mean([p_.get('accum_launches', 1.0) for p_ in phases]))))      # 2 (3 from 2^21 points) when the host scalars go up in chunks that share one reduction: a launch then covers n / launches points on average
        alg_bytes = 128.0 * n / launches                       # 32 B scalar + 96 B affine base per point, x the points ONE launch processes
        achieved = alg_bytes / ak / 1e9
        traffic = None
        tf = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(tf):
            try:
                pm = json.load(open(tf))
                # per LAUNCH of k_accum28, like `achieved`: the chunked host-scalar call makes `launches` of them per step (round-4 passes: the mean of the step's launches)
                # (a chunked call of 3 launches — from 2^21 points — has no PMC pass of its own: null rather than a figure for another launch shape)
                traffic = pm.get('round5', pm.get('round4', {})).get('k_accum_hbm_bytes_per_launch_mean') if launches == 2 else (pm.get('k_accum_hbm_bytes_per_launch') if launches == 1 else None)
            except Exception: traffic = None
        wl = ('2^%d-point BLS12-377 G1 Pippenger MSM split over %d GPUs (BASELINE configs[4])' % (total.bit_length() - 1, world)) if strong else \
             ('standalone 2^%d-point BLS12-377 G1 Pippenger MSM (BASELINE configs[1])' % (n.bit_length() - 1))
        out = {
            'metric': 'MSM G1 scalar-muls/sec (2^%d bases%s, BLS12-377, bit-exact)' % ((total.bit_length() - 1, ' total') if strong else (n.bit_length() - 1, ' per GPU')),
            'value': value, 'unit': 'scalar-muls/s', 'n_gpus': world, 'ranks': world, 'rccl_ranks': (dist.get_world_size() if world > 1 and args.backend == 'nccl' else 0),
            'backend': (args.backend if world > 1 else None), 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_step, 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'u32',
            'data': 'synthetic',
            'config': {'workload': wl, 'points_per_gpu': n, 'scalar_distribution': args.scalars,
                       'scalars': 'host memory (pageable), uploaded inside every timed step (SURVEY.md 8d(i)); value_scalars_resident is the variant with scalars already in HBM',
                       'bases': 'pinned in HBM, P_i=(i+1)G generated on device' + ('' if args.no_precompute else '; fixed-base window tables built at pin time (precompute_s, table_bytes)'),
                       'entry_point': 'aleo_mi355x_msm_g1_pinned',
                       # the headline is a FIXED-BASE call (window tables built once per pinned SRS: precompute_s, table_bytes).  The seam the reference has today —
                       # VariableBase::msm(bases, scalars) with nothing pinned — costs what these two say (same points, same scalars, ms per call):
                       'unpinned_seam_ms': {'one_shot_cold (host bases + host scalars, nothing cached: aleo_mi355x_msm_g1)': variants.get('ms_one_shot_cold'),
                                            'pinned_without_tables': variants.get('ms_no_table')},
                       'sharding': ('point-sharded, all-gather of 144-byte partials' if world > 1 else 'single GPU')},
            'precompute_s': precompute_s, 'table_bytes': info['table_bytes'], 'base_row_bytes': info['row_bytes'], 'table_window_bits': info['tier_window_bits'],
            **variants,
            'roofline': {'bound': 'hbm', 'kernel': 'k_accum28 (bucket accumulation, 28-bit limbs)', 'achieved': achieved, 'peak': 8000.0,
                         'unit': 'GB/s', 'frac': achieved / 8000.0, 'traffic': traffic, 'alg_bytes_per_launch': alg_bytes, 'kernel_ms': ak * 1e3,
                         'launches_per_step': launches, 'points_per_launch': n // launches,
                         'note': 'integer-VALU bound by construction (SURVEY.md §8d): %d mixed additions x 10 Fq products per point; '
                                 'measured Fq product peak 81 G/s with 28-bit limbs, 61 G/s with 32-bit limbs (tools/ubench/fq28_mul_bench.hip, fq_mul_bench.hip)' % (16 if args.no_precompute else 13)},
            'phases_ms': {kk: float(np.mean([p_[kk] for p_ in phases])) for kk in phases[0]},
        }
        # the bound this kernel actually runs against (secondary; `roofline` keeps the contract's HBM form): Fq products per second against the
        # micro-benchmarked product peak of the same 28-bit-limb block
        rows_ = 16 if args.no_precompute else 13
        out['roofline']['valu'] = {'fq_products_per_launch': 10.0 * rows_ * n / launches, 'achieved': 10.0 * rows_ * n / launches / ak / 1e9, 'peak': 81.0, 'unit': 'G Fq products/s',
                                   'frac': 10.0 * rows_ * n / launches / ak / 81e9, 'what': '%d windows x 10 products per mixed addition x points; zero digits (1 in 2^20) not subtracted' % rows_}
        if replicas is not None: out['prove_replicas'] = replicas
        if cfg4 is not None: out['config4_2^%d' % args.config4_lg] = cfg4
        if world == 1 and args.concurrent_callers > 1:
            out['concurrent_callers'] = concurrent_callers(aleo_amd, synth, torch, dev, pb, n, args.concurrent_callers)
        if world == 1 and not args.no_kzg_chain:
            out['kzg_chain_2^22'] = kzg_chain(aleo_amd, synth, torch, dev)
        if world == 1 and not args.no_variants:
            try: out['msm_g2'] = msm_g2_bench(aleo_amd, synth)
            except SystemExit: raise
            except Exception as e: out['msm_g2'] = {'error': repr(e)[:300]}
        if world == 1 and args.proof_proxy_lg:
            pb.close()
            out['proof_proxy'] = {'2^%d' % lg_: proof_proxy_gpu(aleo_amd, synth, torch, dev, lg_)
                                  for lg_ in sorted({args.proof_proxy_lg, args.proof_proxy_cpu_lg})}
            out['proof_proxy']['schedule'] = PROXY_NOTE
            out['key_synthesis_proxy'] = {'2^%d' % lg_: index_proxy_gpu(aleo_amd, synth, torch, dev, lg_) for lg_ in sorted({args.proof_proxy_lg, args.proof_proxy_cpu_lg})}
            out['key_synthesis_proxy']['schedule'] = INDEX_NOTE
            pb = aleo_amd.PinnedBases.generate_multiples(gen, first, n)
        if world == 1 and args.varuna_lg:
            pb.close()
            out['varuna_prove'] = varuna_prove(synth, torch, args.varuna_lg)
            if args.varuna_big_lg and args.varuna_big_lg != args.varuna_lg:
                try: out['varuna_prove_2^%d' % args.varuna_big_lg] = varuna_prove_big(synth, args.varuna_big_lg)
                except Exception as e: out['varuna_prove_2^%d' % args.varuna_big_lg] = {'error': repr(e)[:300]}      # secondary: never costs the headline line
            pb = aleo_amd.PinnedBases.generate_multiples(gen, first, n)
        if world == 1 and 'varuna_prove' in out:
            # SURVEY.md §8d(ii) in one place: constraints/s of real proofs (one circuit; eight instances; several circuits as Trace::prove_execution batches them)
            vp = out['varuna_prove']; big = out.get('varuna_prove_2^%d' % args.varuna_big_lg, {})
            summ = {'2^%d' % args.varuna_lg: vp.get('constraints_per_s'), '8 x 2^%d' % args.varuna_lg: vp.get('instances_8', {}).get('constraints_per_s'),
                    'several_circuits': vp.get('several_circuits', {}).get('constraints_per_s')}
            if 'constraints_per_s' in big:
                summ['2^%d' % args.varuna_big_lg] = big['constraints_per_s']; summ['8 x 2^%d' % args.varuna_big_lg] = big.get('instances_8', {}).get('constraints_per_s')
                bh = big.get('bit_heavy_witness_commit_lagrange', {})
                if 'instances_8' in bh: summ['8 x 2^%d bit-heavy, commit_lagrange' % args.varuna_big_lg] = bh['instances_8'].get('constraints_per_s')
            out['prove_constraints_per_s'] = summ
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, pb, scalars, aleo_amd)
        print(json.dumps(out), flush=True)
    if world > 1:
        # Secondary, outside the timed region and AFTER the MSM line is out: the sharded (4-step) NTT with its one RCCL all-to-all.
        # It must never cost the headline run, so it runs in a CHILD process per rank (own process group on the next port): the
        # parent has already printed its line and exits 0 whatever happens to the probe; a child that hangs is killed after 150 s
        # and reported on stderr, never as success.
        sys.stdout.flush()
        dist.barrier(); dist.destroy_process_group()
        if args.sharded_ntt_lg:
            import subprocess
            env = dict(os.environ, MASTER_PORT=str(int(os.environ.get('MASTER_PORT', '29500')) + 1), MASTER_ADDR=os.environ.get('MASTER_ADDR', '127.0.0.1'))
            env.pop('TORCHELASTIC_USE_AGENT_STORE', None)          # the children rendezvous on their own store (rank 0 hosts it), not the launcher's
            cmd = [sys.executable, os.path.abspath(__file__), '--aux-sharded-ntt', '--gpus', str(world), '--sharded-ntt-lg', str(args.sharded_ntt_lg), '--backend', args.backend]
            try:
                r = subprocess.run(cmd, env=env, timeout=150, capture_output=True, text=True)
                if rank == 0:
                    line = [l for l in r.stderr.splitlines() if l.startswith('{"aux"')]
                    print(line[-1] if line else json.dumps({'aux': 'sharded_ntt', 'error': 'probe exited with %d' % r.returncode, 'stderr_tail': r.stderr[-400:]}), file=sys.stderr, flush=True)
            except subprocess.TimeoutExpired:
                print(json.dumps({'aux': 'sharded_ntt', 'error': 'timeout: the probe did not finish within 150 s (child killed)', 'rank': rank}), file=sys.stderr, flush=True)


def config4_split(aleo_amd, synth, adist, dist, torch, rank, world, gather_dev, barrier, lg_total, backend, steps=3, no_precompute=False):
    """BASELINE configs[4] inside the default multi-rank run: one MSM of 2^lg_total points, rank r owning the contiguous shard adist.shard_range gives it
    (P_i = (i + 1) G generated in its HBM, fixed-base tables built), host scalars uploaded inside every timed step, the 144-byte partials all-gathered
    (RCCL when backend is nccl) and added in rank order.  Every rank runs it (collectives inside); returns the dict rank 0 reports: whole-MSM time = max over
    the ranks of the barrier-to-barrier time, each rank's own time per step, and the result gate (k G in big integers)."""
    import numpy as np
    total = 1 << lg_total
    lo, hi = adist.shard_range(total, rank, world); n = hi - lo
    if n * 13 >= (1 << 32): return {'error': '%d points per rank exceed one launch chain (2^32 (bucket, point) pairs): more ranks needed' % n}
    pb4 = aleo_amd.PinnedBases.generate_multiples(synth.generator_affine104(), lo + 1, n)
    try:
        if not no_precompute: pb4.precompute()
        sc = synth.uniform_scalars(n, 0xA1E00004 + rank)                   # seed 0xA1E00000 + config id (+ rank: every shard its own stream)
        gather = adist.PartialGather(world, gather_dev)
        def step(): return aleo_amd.g1_sum(gather(aleo_amd.VariableBase.msm(pb4, sc)))
        res = step(); mine = []
        barrier(); t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter(); part = aleo_amd.VariableBase.msm(pb4, sc); mine.append(time.perf_counter() - t1)
            res = aleo_amd.g1_sum(gather(part))
        barrier(); el = time.perf_counter() - t0
        t = torch.tensor([el, float(np.mean(mine))], dtype=torch.float64, device=gather_dev if gather_dev is not None else 'cpu')
        per = torch.zeros((world, 2), dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(per.view(-1), t)
        per = per.cpu().numpy(); el_max = float(per[:, 0].max())
        k = synth.weighted_scalar_sum(sc, lo + 1)
        ks = gather(synth.int_to_limbs(k, 4)); k = sum(synth.limbs_to_int(row) for row in ks) % synth.FR_MODULUS
        ok = bool(result_is_multiple_of_generator(synth, res, k))
        return {'workload': '2^%d-point BLS12-377 G1 Pippenger MSM split over %d ranks (BASELINE configs[4]), host scalars uploaded inside the step' % (lg_total, world),
                'ms': el_max / steps * 1e3, 'scalar_muls_per_s': total * steps / el_max, 'per_rank_msm_ms': [float(x) * 1e3 for x in per[:, 1]], 'points_per_rank': n,
                'steps': steps, 'scaling': 'strong', 'rccl_ranks': (dist.get_world_size() if backend == 'nccl' else 0), 'backend': backend,
                'result_is_k_times_generator': ok}
    finally:
        pb4.close()


def spawn_ranks(n: int) -> int:
    """The launcher half of `python bench.py --gpus N`: N child processes of this same command line, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set as torch.distributed.run would set them (rank 0 hosts the store on a free port of 127.0.0.1).  Rank 0's stdout is
    relayed line by line (the ONE JSON line); the other ranks' stdout goes to stderr.  If a rank fails the others are ended (by their own PIDs) and
    its status is returned; ALEO_BENCH_LAUNCH_TIMEOUT (seconds, default 3000) bounds the whole run."""
    import socket, subprocess, threading
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line); sys.stdout.flush()
    t = threading.Thread(target=relay, daemon=True); t.start()
    deadline = time.time() + float(os.environ.get('ALEO_BENCH_LAUNCH_TIMEOUT', '3000'))
    status = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or all(c is not None for c in codes) or time.time() > deadline:
            if bad: status = bad[0]
            elif any(c is None for c in codes): status = 124; print('bench: launcher timeout', file=sys.stderr)
            break
        time.sleep(0.2)
    for p in procs:                                            # only reached with live children after a failure / timeout
        if p.poll() is None:
            p.terminate()
            try: p.wait(timeout=20)
            except subprocess.TimeoutExpired: p.kill(); p.wait()
    t.join(timeout=5)
    return status


def launch_check(args) -> int:
    """bench.py --gpus N --backend gloo --launch-check: everything of the N > 1 path that is not arithmetic — the ranks meet, all-gather a 144-byte partial
    each (PartialGather: the exchange of a point-sharded MSM), add nothing, reduce the elapsed time with MAX, and rank 0 prints a line with
    n_gpus = N and value null.  Runs on a CPU box (tests/test_dist_cpu.py); it measures nothing."""
    import torch
    import torch.distributed as dist
    from aleo_amd import dist as adist
    rank = int(os.environ.get('RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.backend != 'gloo': raise SystemExit('--launch-check is the CPU rehearsal: use --backend gloo')
    if world > 1: dist.init_process_group('gloo', rank=rank, world_size=world)
    gather = adist.PartialGather(world, None) if world > 1 else None
    part = np.full(18, rank + 1, dtype=np.uint64)
    if world > 1: dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows = gather(part) if gather else part.reshape(1, 18)
    if world > 1: dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1: dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ok = [int(r[0]) for r in np.asarray(rows).reshape(world, 18)] == list(range(1, world + 1))
    # the extra exchange of config4_split (the per-rank times of the 2^26-point MSM the default N > 1 run also reports): an all-gather of two doubles per rank
    per = torch.zeros((world, 2), dtype=torch.float64)
    if world > 1: dist.all_gather_into_tensor(per.view(-1), torch.tensor([float(rank), 1.0], dtype=torch.float64))
    cfg4_ok = world == 1 or [float(x) for x in per[:, 0]] == [float(r) for r in range(world)]
    ok = ok and cfg4_ok
    if rank == 0:
        print(json.dumps({'metric': 'launch check (no arithmetic)', 'value': None, 'unit': 'scalar-muls/s', 'n_gpus': world, 'ranks': dist.get_world_size() if world > 1 else 1,
                          'rccl_ranks': 0, 'backend': args.backend, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': float(el.item()) / max(args.steps, 1) * 1e3,
                          'gather_in_rank_order': ok, 'config4_per_rank_gather': cfg4_ok, 'launch_check': True}), flush=True)
    if world > 1: dist.barrier(); dist.destroy_process_group()
    return 0 if ok else 1


def _g1_times(P, k, q):
    """k * P on y^2 = x^3 + 1 over Fq in plain Python integers (Jacobian double-and-add) — the gate's own arithmetic: bench.py's correctness check must
    not lean on the library it measures, and oracle/ is only for the cpu_baseline leg."""
    X, Y, Z = 1, 1, 0
    for bit in bin(k)[2:] if k else '':
        if Z:                                                   # double (a = 0)
            A = X * X % q; B = Y * Y % q; C = B * B % q; D = 2 * ((X + B) * (X + B) - A - C) % q; E = 3 * A % q
            X3 = (E * E - 2 * D) % q; Z = 2 * Y * Z % q; Y = (E * (D - X3) - 8 * C) % q; X = X3
        if bit == '1':
            if not Z: X, Y, Z = P[0], P[1], 1
            else:                                               # mixed addition; P never equals +-acc here (k < r, P of order r) except through the doubling above
                ZZ = Z * Z % q; U2 = P[0] * ZZ % q; S2 = P[1] * Z % q * ZZ % q; H = (U2 - X) % q; R_ = (S2 - Y) % q
                HH = H * H % q; HHH = H * HH % q; V = X * HH % q
                X3 = (R_ * R_ - HHH - 2 * V) % q; Y = (R_ * (V - X3) - Y * HHH) % q; Z = Z * H % q; X = X3
    if not Z: return None
    zi = pow(Z, -1, q); return (X * zi * zi % q, Y * zi * zi % q * zi % q)


def result_is_multiple_of_generator(synth, res, k) -> bool:
    """res: uint64[18] Jacobian (x, y, 1) in Montgomery form, or (1, 1, 0); k: the expected discrete logarithm."""
    q = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
    rinv = pow(1 << 384, -1, q)
    lim = lambda a: sum(int(v) << (64 * i) for i, v in enumerate(a))
    r = np.asarray(res, dtype=np.uint64).reshape(18)
    got = None if not r[12:].any() else (lim(r[:6]) * rinv % q, lim(r[6:12]) * rinv % q)
    g = np.ascontiguousarray(synth.generator_affine104()[:96]).view(np.uint64)
    G = (lim(g[:6]) * rinv % q, lim(g[6:12]) * rinv % q)
    return got == _g1_times(G, k % synth.FR_MODULUS, q)


def in_process_sharded(args):
    """bench.py --config 4 --in-process --gpus N: BASELINE configs[4] driven by ONE process over N device contexts of the C ABI
    (aleo_mi355x_bases_generate_sharded / aleo_mi355x_msm_g1_sharded: contiguous point shards, per-device Pippenger from one host thread per shard,
    the 144-byte partials added on the host in shard order — no collective).  With fewer visible devices than N the devices are listed cyclically
    (several shards share a card: a rehearsal of the path, not a scaling measurement; the line says which)."""
    import torch
    import aleo_amd
    from aleo_amd import synth
    if not torch.cuda.is_available(): raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    L = aleo_amd.lib(); aleo_amd._lib.check(L.aleo_mi355x_init(0), 'init')
    vis = torch.cuda.device_count(); G = max(1, args.gpus)
    devices = [g % vis for g in range(G)]
    total = 1 << (args.lg_n if args.lg_n is not None else 26)
    if (total // G + 1) * 13 >= (1 << 32): raise SystemExit('bench --config 4: the shards exceed one launch; use more devices or --lg-n')
    gen = synth.generator_affine104()
    t0 = time.perf_counter()
    sb = aleo_amd.ShardedBases.generate_multiples(gen, 1, total, devices=devices, precompute=not args.no_precompute)
    setup_s = time.perf_counter() - t0
    scalars = (synth.uniform_scalars if args.scalars == 'uniform' else synth.witness_like_scalars)(total, 0xA1E00002)
    for _ in range(args.warmup): res = aleo_amd.VariableBase.msm_sharded(sb, scalars)
    t0 = time.perf_counter()
    for _ in range(args.steps): res = aleo_amd.VariableBase.msm_sharded(sb, scalars)
    elapsed = time.perf_counter() - t0
    if not result_is_multiple_of_generator(synth, res, synth.weighted_scalar_sum(scalars, 1)):
        raise SystemExit('bench: sharded MSM result does not equal k*G — refusing to report a number for a wrong result')
    print(json.dumps({
        'metric': 'MSM G1 scalar-muls/sec (2^%d bases total, BLS12-377, bit-exact)' % (total.bit_length() - 1), 'value': total * args.steps / elapsed, 'unit': 'scalar-muls/s',
        'n_gpus': len(set(devices)), 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'u32', 'data': 'synthetic',
        'config': {'workload': '2^%d-point BLS12-377 G1 Pippenger MSM split into %d shards over %d device(s) of one process (BASELINE configs[4])' % (total.bit_length() - 1, G, len(set(devices))),
                   'shards': sb.shards(), 'entry_point': 'aleo_mi355x_msm_g1_sharded', 'scalars': 'host memory, uploaded inside every timed step',
                   'sharding': 'one process, one host thread and device context per shard; partial sums added on the host in shard order (no collective)',
                   'rehearsal_on_shared_devices': len(set(devices)) < G},
        'setup_s': setup_s}), flush=True)
    sb.close()
    return 0


def aux_sharded_ntt(args):
    """Child-process entry (bench.py --aux-sharded-ntt): only the sharded-NTT probe, on its own process group."""
    import torch
    import torch.distributed as dist
    import aleo_amd
    from aleo_amd import synth
    from aleo_amd import dist as adist
    rank = int(os.environ.get('RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1')); local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dev_index = local_rank % torch.cuda.device_count(); torch.cuda.set_device(dev_index); dev = torch.device('cuda', dev_index)
    if args.backend == 'nccl': dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else: dist.init_process_group('gloo', rank=rank, world_size=world)
    aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_init_device(dev_index), 'init')
    try:
        sn = sharded_ntt_probe(aleo_amd, adist, synth, torch, dist, dev, rank, world, args.sharded_ntt_lg)
    except Exception as e:      # noqa: BLE001
        sn = {'error': repr(e)}
    if rank == 0:
        print(json.dumps({'aux': 'sharded_ntt', **sn}), file=sys.stderr, flush=True)
    dist.barrier(); dist.destroy_process_group()
    return 1 if 'error' in sn else 0


def concurrent_callers(aleo_amd, synth, torch, dev, pb, n, T, reps=8):
    """Secondary: T host threads (snarkVM's rayon commitments of one round) each running MSMs of the headline size against
    the same pinned set; every call takes its own library slot, so sort / reduce phases of one call overlap another's
    accumulation.  Aggregate rate, NOT the headline `value` (which has one call in flight)."""
    import threading
    ds = [torch.from_numpy(synth.uniform_scalars(n, 0xA1E00030 + t).view(np.int64)).to(dev) for t in range(T)]
    torch.cuda.synchronize()

    def work(t):
        torch.cuda.set_device(dev)
        for _ in range(reps): aleo_amd.VariableBase.msm_device(pb, ds[t].data_ptr(), n)
    dt = None
    for _ in range(2):                       # first round warms the slots' workspaces
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        dt = time.perf_counter() - t0
    return {'threads': T, 'ms_per_msm_aggregate': dt / (T * reps) * 1e3, 'scalar_muls_per_s': T * reps * n / dt}


def sharded_ntt_probe(aleo_amd, adist, synth, torch, dist, dev, rank, world, lg_n, reps=5):
    """2^lg_n-point NTT spread over the ranks (aleo_amd.dist.ShardedDomain): local column transforms, twiddle, ONE
    all-to-all, local row transforms.  Each rank checks its evaluation block against a single-GPU transform of the same
    vector, then the forward transform is timed (max over ranks)."""
    if world & (world - 1) or lg_n < 2 * (world.bit_length() - 1): return {'skipped': 'needs a power-of-two world'}
    n = 1 << lg_n
    x = synth.uniform_scalars(n, 0xA1E00020)                                  # canonical < r, read as Montgomery
    dom = adist.ShardedDomain(lg_n, rank, world)
    mine = torch.from_numpy(dom.coefficient_shard(x).view(np.int64).copy()).to(dev)
    full = torch.from_numpy(x.view(np.int64).copy()).to(dev)
    aleo_amd.EvaluationDomain(n).ntt_device(full.data_ptr(), 0, 0, 0, adist._torch_stream_handle())
    ev = dom.forward(mine.clone()); torch.cuda.synchronize()
    idx = torch.from_numpy(dom.evaluation_indices()).to(dev)
    ok = bool((ev == full[idx]).all())
    back_ok = bool((dom.inverse(ev.clone()) == mine).all())
    del full, idx
    dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        ev = dom.forward(mine.clone())
    torch.cuda.synchronize(); dist.barrier(); dt = (time.perf_counter() - t0) / reps
    t = torch.tensor([dt, 0.0 if (ok and back_ok) else 1.0], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return {'lg_n': lg_n, 'ranks': world, 'ms': float(t[0]) * 1e3, 'alg_GBps_aggregate': 64.0 * n / float(t[0]) / 1e9,
            'matches_single_gpu_transform_on_all_ranks': float(t[1]) == 0.0}


def msm_g2_bench(aleo_amd, synth):
    """VariableBase::msm::<G2Affine> (SRS / setup paths, never in the prover's loop) through the one-shot entry point — host bases
    (200-byte snarkVM rows) and host scalars, nothing resident — at 2^16 and 2^20 points; results gated against (sum s_i w_i) G2 in Python integers.
    fq_product_equivalents_per_s: 30 Fq products per mixed addition (ten Fq2 products) x points x windows / WALL time of the call — a lower bound of what
    the accumulation kernel sustains (the 81 G/s of roofline.valu is the G1 block's peak)."""
    from aleo_amd import msm as M
    out = {'entry_point': 'aleo_mi355x_msm_g2', 'bases': 'P_i = ((i mod 4096) + 1) G2, host memory, uploaded and unpacked inside every call',
           'before_round4': {'2^16_ms': 7.99, '2^20_ms': 95.27, 'file': 'profiles/r04_g2_round2_kernel.jsonl', 'what': 'round-3 state measured at the start of round 4: one-lane 32-bit kernels, 200-byte rows repacked by a host loop'},
           'round2_kernels_with_device_unpack': {'2^16_ms': 7.28, '2^20_ms': 30.61, 'file': 'profiles/r04_g2_round2_kernels_device_unpack.jsonl', 'what': 'ALEO_MI355X_G2_PAIR28=0 on the same box as profiles/r04_g2_pair28.jsonl'}}
    for lg in (16, 20):
        n = 1 << lg
        B = synth.g2_multiples_affine200(n); S = synth.uniform_scalars(n, 0xA1E00077)
        res = M.msm_g2(B, S)
        if not synth.g2_result_gate(res, S): raise SystemExit('bench: G2 MSM result differs from its discrete logarithm times the generator')
        reps = 3 if lg >= 18 else 6; t0 = time.perf_counter()
        for _ in range(reps): M.msm_g2(B, S)
        dt = (time.perf_counter() - t0) / reps
        c = min(16, lg - 4); W = (254 + c - 1) // c
        out['2^%d' % lg] = {'ms': dt * 1e3, 'scalar_muls_per_s': n / dt, 'windows': W, 'fq_product_equivalents_per_s': 30.0 * n * W / dt}
        with M.PinnedG2Bases(B) as pg:                      # the same request against a set resident in HBM (aleo_mi355x_msm_g2_pinned: only the scalars move)
            if not (pg.msm(S) == res).all(): raise SystemExit('bench: the pinned G2 set and the one-shot call disagree')
            t0 = time.perf_counter()
            for _ in range(reps): pg.msm(S)
            dp = (time.perf_counter() - t0) / reps
        out['2^%d' % lg].update({'pinned_ms': dp * 1e3, 'pinned_scalar_muls_per_s': n / dp, 'pinned_fq_product_equivalents_per_s': 30.0 * n * W / dp})
    return out


def kzg_chain(aleo_amd, synth, torch, dev, lg=22, reps=5):
    """BASELINE configs[2]: 2^22 scalar-field iNTT followed by a 2^22 G1 MSM over the result without leaving HBM
    (the KZG10::commit shape).  Secondary measurement: NTT time by HIP events on the launch stream, algorithmic bytes
    64 B per element; the commit through the library's own phase timers."""
    n = 1 << lg
    pb = aleo_amd.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n).precompute()
    evals = torch.from_numpy(synth.uniform_scalars(n, 0xA1E00003).view(np.int64)).to(dev)      # canonical < r, read as Montgomery
    d = aleo_amd.EvaluationDomain(n)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        buf = evals.clone()
        d.ntt_device(buf.data_ptr(), 0, 1, 0, st.cuda_stream); st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            d.ntt_device(buf.data_ptr(), 0, 1, 0, st.cuda_stream)
        e1.record(st); st.synchronize()
        ntt_ms = e0.elapsed_time(e1) / reps
        buf.copy_(evals); d.ntt_device(buf.data_ptr(), 0, 1, 0, st.cuda_stream); st.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        cm = aleo_amd.KZG10.commit_device(pb, buf.data_ptr(), n)
    commit_ms = (time.perf_counter() - t0) / reps * 1e3
    pb.close()
    gbps = 64.0 * n / (ntt_ms * 1e-3) / 1e9
    return {'ntt_ms': ntt_ms, 'ntt_alg_GBps': gbps, 'ntt_frac_hbm': gbps / 8000.0, 'commit_ms': commit_ms,
            'commit_scalar_muls_per_s': n / (commit_ms * 1e-3), 'chain_ms': ntt_ms + commit_ms,
            'commitment_x_limb0': int(np.frombuffer(cm.tobytes()[:8], dtype=np.uint64)[0])}


PROXY_NOTE = ('operator-level proxy for constraints/s (SURVEY.md §8d): the MSM/NTT/field-op schedule of one single-instance Varuna proof with '
              '|H| = |K| = 2^k, replayed on synthetic vectors of those sizes [schedule UPSTREAM-RECALL, see DESIGN.md §4c]; the commitments of a '
              'round go through ONE batched call (aleo_mi355x_kzg_commit_batch_device), the two openings divide by (X - z) on the device; not a proof')


VARUNA_TAU, VARUNA_S = 0x1F3A9C0D5E7B24681357ACE02468BDF013579BDF02468ACE1234567, 0x0FEDCBA9876543210123456789ABCDEF55AA
VARUNA_NOTE = ('aleo_mi355x_varuna_prove (one C call per proof): the four AHP rounds, evaluations and both KZG openings of one Marlin/Varuna-shaped proof for a synthetic satisfiable R1CS '
               '(every constraint multiplies two short linear combinations; a few wide rows), all circuit-sized work on the device through the C ABI; '
               'Poseidon-over-Fq transcript (upstream construction; absorb order recalled) and synthetic SRS, so proofs are checked by the restatement in oracle/varuna_ref.py, not by snarkVM (DESIGN.md)')


def _varuna_instance(synth, lg, seed, bits=False, lagrange=False):
    from aleo_amd import varuna
    n = (1 << lg) - 64 if lg >= 8 else (1 << lg) - 4
    csr, z = synth.synthetic_r1cs_bits(n, 4, seed) if bits else synth.synthetic_r1cs(n, 4, seed, long_rows=4 if lg >= 10 else 1)
    zz = np.stack([synth.int_to_limbs(v, 4) for v in z])
    nnz = max(int(csr[m][0][-1]) for m in 'abc'); n_k = 2
    while n_k < nnz: n_k *= 2
    n_h = 1
    while n_h < max(n, 4 + len(z) - 4, 8): n_h *= 2
    D = 1
    while D < max(3 * n_h, n_k): D *= 2
    ck = varuna.synthetic_committer_key(VARUNA_TAU, VARUNA_S, D - 1, lagrange_size=n_h if lagrange else 0, range_window=(16 if n_h >= (1 << 18) else 13) if lagrange else 0)
    return n, csr, z, zz, ck, D - 1


def varuna_prove(synth, torch, lg, reps=7, in_flight=4):
    """constraints/s of the prover above the operators: one proof at a time, and `in_flight` proofs of the
    same circuit from as many host threads, each on its own stream."""
    import threading
    from aleo_amd import varuna
    n, csr, z, zz, ck, D = _varuna_instance(synth, lg, 40 + lg)
    try:
        ix = varuna.CircuitIndex(csr, n, 4, len(z) - 4, ck)
        varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck).close()
        t0 = time.perf_counter(); nx = varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck); index_s = time.perf_counter() - t0      # key synthesis: one call
        if nx.vk_bytes != ix.vk_bytes or nx.prove(zz, 1) != varuna.prove_native(ix, zz, 1): raise SystemExit('bench: the two index builders disagree')
        nx.close()
        ts = []
        for rep in range(reps + 2):                        # one call of the C ABI per proof (aleo_mi355x_varuna_prove)
            t = time.perf_counter(); data = varuna.prove_native(ix, zz, 1000 + rep); ts.append((time.perf_counter() - t) * 1e3)
        med = float(np.median(ts[2:])); rounds = varuna.native_timing()
        tp = []
        for rep in range(5):                               # the same sequence driven from Python through the public entry points
            t = time.perf_counter(); pr = varuna.prove(ix, zz, 1000 + rep); tp.append((time.perf_counter() - t) * 1e3)
        if pr.to_bytes() != varuna.prove_native(ix, zz, 1004): raise SystemExit('bench: the two host sides of the prover disagree')
        tb = []
        for rep in range(5):                               # prove_batch shape: four instances of the circuit in one proof
            t = time.perf_counter(); datab = varuna.prove_native(ix, [zz] * 4, 2000 + rep); tb.append((time.perf_counter() - t) * 1e3)
        mb = float(np.median(tb[1:]))
        tb8 = []
        for rep in range(4):
            t = time.perf_counter(); varuna.prove_native(ix, [zz] * 8, 3000 + rep); tb8.append((time.perf_counter() - t) * 1e3)
        mb8 = float(np.median(tb8[1:]))
        per = 6
        def work(k):
            for rep in range(per): varuna.prove_native(ix, zz, 5000 + 100 * k + rep)
        for _ in range(2):
            th = [threading.Thread(target=work, args=(k,)) for k in range(in_flight)]
            t = time.perf_counter()
            for x in th: x.start()
            for x in th: x.join()
            dt = time.perf_counter() - t
        lock = {}
        try:                                               # eight independent proofs per aleo_mi355x_varuna_prove_many call (every round's commitments in one launch chain)
            with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx8:
                reqs = [([nx8], [[zz]], 7000 + q) for q in range(8)]
                got = varuna.prove_many_native(reqs)
                if got[3] != varuna.prove_native(ix, zz, 7003): raise SystemExit('bench: a lockstep proof differs from the single call')
                tl = []
                for rep in range(4):
                    t = time.perf_counter(); varuna.prove_many_native(reqs); tl.append(time.perf_counter() - t)
                ml = float(np.median(tl[1:]))
                lock = {'proofs_per_call': 8, 'ms_per_call': ml * 1e3, 'proofs_per_s': 8 / ml, 'constraints_per_s': 8 * n / ml, 'entry_point': 'aleo_mi355x_varuna_prove_many',
                        'what': 'independent proofs (own seed, transcript, output), byte-equal to the single calls; one host thread'}
                def two(k):                                # two such calls in flight (what aleo_mi355x::ProvingQueue does with its two drainer threads)
                    rq = [([nx8], [[zz]], 7100 + 10 * k + q) for q in range(8)]
                    for rep in range(3): varuna.prove_many_native(rq)
                for _ in range(2):
                    th = [threading.Thread(target=two, args=(k,)) for k in range(2)]
                    t = time.perf_counter()
                    for x in th: x.start()
                    for x in th: x.join()
                    d2 = time.perf_counter() - t
                lock['two_callers'] = {'proofs_per_s': 2 * 3 * 8 / d2, 'constraints_per_s': n * 2 * 3 * 8 / d2}
        except SystemExit: raise
        except Exception as e: lock = {'error': repr(e)[:300]}
        try: several = varuna_prove_several(synth, ck, lg)
        except Exception as e: several = {'error': repr(e)[:300]}
        try: sweep = density_sweep(synth, lg)
        except Exception as e: sweep = {'error': repr(e)[:300]}
        return {'constraints': n, 'domain_h': ix.n_h, 'domain_k': ix.n_k, 'max_degree': D, 'index_s': index_s, 'prove_ms': med, 'constraints_per_s': n / med * 1e3,
                'rounds_ms': rounds, 'proof_bytes': len(data), 'entry_point': 'aleo_mi355x_varuna_prove', 'several_circuits': several, 'density_sweep': sweep,
                'circuit': 'synthetic_r1cs: 2.02 / 1.5 / 1 non-zeros per row of A / B / C (every constraint multiplies two short combinations into a new variable); density_sweep holds denser ones',
                'python_host_ms': float(np.median(tp[1:])),
                'instances_4': {'prove_ms': mb, 'constraints_per_s': 4 * n / mb * 1e3, 'proof_bytes': len(datab)},
                'instances_8': {'prove_ms': mb8, 'constraints_per_s': 8 * n / mb8 * 1e3},
                'in_flight_%d' % in_flight: {'proofs_per_s': in_flight * per / dt, 'constraints_per_s': n * in_flight * per / dt}, 'lockstep_8': lock, 'what': VARUNA_NOTE}
    finally:
        ck.close()


def density_sweep(synth, lg, reps=5):
    """constraints/s against circuit density (tools/density_sweep.py holds the full sweep, profiles/r03_density_sweep.jsonl its 2^15 / 2^18 results):
    the headline circuit has ~2 non-zeros per row; programs built from hash gadgets have denser A / B rows, hence larger non-zero domains |K_M| and
    more MSM points per constraint.  Here: 8 and 16 non-zeros per row of A and B, and a Poseidon-gadget-shaped circuit (dense MDS rows, x^17 chains)."""
    from aleo_amd import varuna
    n = (1 << lg) - 64; out = []
    for name, make in (('density 8 / 8', lambda: synth.synthetic_r1cs_density(n, 4, 908, 8, 8)), ('density 16 / 16', lambda: synth.synthetic_r1cs_density(n, 4, 916, 16, 16)),
                       ('poseidon-shaped, width 9', lambda: synth.synthetic_r1cs_poseidon(n, 4, 77, 9)),
                       ('hash_psd2 chain: the real Poseidon gadget (rate 2), %d hashes' % (n // 276), lambda: synth.poseidon_chain_r1cs(n // 276, 79)[:2])):
        csr, z = make(); zz = np.stack([synth.int_to_limbs(v, 4) for v in z])
        n_c = len(csr['a'][0]) - 1; n_pub = 2 if 'hash_psd2' in name else 4
        nnz = [int(csr[m][0][-1]) for m in 'abc']; n_k = 2
        while n_k < max(nnz): n_k *= 2
        D = 1
        while D < max(3 << lg, n_k): D *= 2
        ck = varuna.synthetic_committer_key(0x1F3A9C0D5E7B24681357ACE02468BDF013579BDF02468ACE1234567, 0x0FEDCBA9876543210123456789ABCDEF55AA, D - 1)
        try:
            with varuna.NativeCircuitIndex(csr, n_c, n_pub, len(z) - n_pub, ck) as nx:
                ts = []
                for r_ in range(reps + 2):
                    t = time.perf_counter(); nx.prove(zz, 100 + r_); ts.append((time.perf_counter() - t) * 1e3)
                ms = float(np.median(ts[2:])); km = nx.n_k_m; nh = nx.n_h
                pts = 3 * (nh + 1) + 3 * nh + (nh - 1) + 2 * nh + sum(k - 1 for k in km) + max(km) + (3 * nh - 1) + (max(km) - 1)
                out.append({'circuit': name, 'constraints': n_c, 'nnz_per_row': [round(v / n_c, 2) for v in nnz], 'n_k': km, 'msm_points_per_constraint': round(pts / n_c, 1), 'prove_ms': ms, 'constraints_per_s': n_c / ms * 1e3})
        finally:
            ck.close()
    return out


def prove_replicas(synth, dist, lg, world, gather_dev, barrier, reps=6):
    """N > 1, secondary (SURVEY.md §8e "independent proofs: replicas"): every rank pins its own committer key and index on its GPU and proves the same
    2^lg-constraint circuit `reps` times between barriers — no data-path collective; aggregate proofs/s and constraints/s from the slowest rank's time."""
    import torch
    from aleo_amd import varuna
    dt, n, err, met = -1.0, 0, None, 0
    try:
        n, csr, z, zz, ck, D = _varuna_instance(synth, lg, 40 + lg)
        try:
            with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx:
                nx.prove(zz, 1); nx.prove(zz, 2)
                barrier(); met = 1; t0 = time.perf_counter()
                for rep in range(reps): nx.prove(zz, 10 + rep)
                barrier(); met = 2; dt = time.perf_counter() - t0
        finally:
            ck.close()
    except Exception as e:                                  # a rank that failed would leave the others waiting at a barrier: it still meets the ones it has not reached
        err = repr(e)[:200]; dt = -1.0
        try:
            for _ in range(2 - met): barrier()
        except Exception: pass
    t = torch.tensor([dt, 1.0 if err is None else 0.0], dtype=torch.float64, device=gather_dev if gather_dev is not None else 'cpu')
    worst = t.clone(); dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    ok = t.clone(); dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok[1].item()) < 1.0: return {'error': err or 'another rank failed'}
    el = float(worst[0].item())
    return {'constraints': n, 'proofs_per_rank': reps, 'proofs_per_s': world * reps / el, 'constraints_per_s': world * reps * n / el, 'sharding': 'replicas: one proof per GPU at a time, no collective'}


def varuna_prove_several(synth, ck, lg):
    """One proof over several circuits (Varuna::prove_batch with a map of proving keys — the shape of a transaction: a function, a second function
    called twice, a fee): circuits of 2^lg, 2^(lg-1) (two instances) and 2^(lg-2) constraints against the committer key of the largest, proved in ONE
    call (aleo_mi355x_varuna_prove_batch_indexed) and, for comparison, as three separate proofs."""
    from aleo_amd import varuna
    shapes = [(lg, 1), (lg - 1, 2), (lg - 2, 1)]
    nx, za, total = [], [], 0
    try:
        for j, (l, k) in enumerate(shapes):
            n = (1 << l) - 64
            csr, z = synth.synthetic_r1cs(n, 4, 140 + 7 * j + l, long_rows=4)
            nx.append(varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck))
            za.append([np.stack([synth.int_to_limbs(v, 4) for v in z])] * k); total += n * k
        varuna.prove_batch_native(nx, za, 1)
        tb = []
        for rep in range(6):
            t = time.perf_counter(); data = varuna.prove_batch_native(nx, za, 50 + rep); tb.append((time.perf_counter() - t) * 1e3)
        ts = []
        for rep in range(6):
            t = time.perf_counter()
            for x, zz in zip(nx, za): x.prove(zz, 70 + rep)
            ts.append((time.perf_counter() - t) * 1e3)
        mb, ms = float(np.median(tb[1:])), float(np.median(ts[1:]))
        return {'circuits': [{'constraints': (1 << l) - 64, 'instances': k} for l, k in shapes], 'constraints': total, 'one_proof_ms': mb, 'constraints_per_s': total / mb * 1e3,
                'proof_bytes': len(data), 'three_separate_proofs_ms': ms, 'entry_point': 'aleo_mi355x_varuna_prove_batch_indexed'}
    finally:
        for x in nx: x.close()


_BIG_PROOFS = []       # device proofs of the large circuit, verified in the cpu_baseline leg (the only place bench.py may use oracle/)


def _varuna_big_once(synth, lg, bits, lagrange):
    from aleo_amd import varuna
    t0 = time.perf_counter(); n, csr, z, zz, ck, D = _varuna_instance(synth, lg, 40 + lg, bits, lagrange); prep_s = time.perf_counter() - t0
    try:
        t0 = time.perf_counter(); nx = varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck); index_s = time.perf_counter() - t0
        try:
            nx.prove(zz, 1)
            t1 = []
            for rep in range(3):
                t = time.perf_counter(); nx.prove(zz, 10 + rep); t1.append((time.perf_counter() - t) * 1e3)
            rounds = varuna.native_timing()
            t8 = []
            for rep in range(4):
                t = time.perf_counter(); data = nx.prove([zz] * 8, 20 + rep); t8.append((time.perf_counter() - t) * 1e3)
            m1, m8 = float(np.median(t1)), float(np.median(t8[1:]))
            if not bits: _BIG_PROOFS.append({'constraints': n, 'vk': nx.vk_bytes, 'public': [int(v) for v in z[:4]], 'proof': nx.prove(zz, 12), 'proof_8': data, 'max_degree': D})      # for the cpu_baseline leg's verifier
            return {'constraints': n, 'domain_h': nx.n_h, 'domains_k': nx.n_k_m, 'max_degree': D, 'index_s': index_s, 'prove_ms': m1, 'constraints_per_s': n / m1 * 1e3,
                    'rounds_ms': rounds, 'instances_8': {'prove_ms': m8, 'constraints_per_s': 8 * n / m8 * 1e3, 'proof_bytes': len(data)},
                    'host_prep_s': prep_s, 'entry_point': 'aleo_mi355x_varuna_index_build + aleo_mi355x_varuna_prove_indexed'}
        finally:
            nx.close()
    finally:
        ck.close()


def varuna_sharded_rehearsal(synth, lg, shard_counts=(2, 8)):
    """SURVEY.md 8 row e2 on ONE card: the 2^lg-constraint proof with a sharded copy of the committer key attached (every commitment of >= 2^16 points cut
    over G shards, each with its own window tables; here all G on device 0, so this times the path — slices, per-shard Pippenger, host merge — not a
    speed-up) against the single-device proof of the same seed: bytes must be equal."""
    import aleo_amd
    from aleo_amd import varuna
    n, csr, z, zz, ck, D = _varuna_instance(synth, lg, 40 + lg)
    out = {'constraints': n, 'min_points_routed_to_shards': 1 << 16, 'devices_visible': 1, 'note': 'all shards on one card: a rehearsal of the multi-device path, not a scaling measurement'}
    try:
        def timed(nx, k):
            nx.prove([zz] * k, 5); ts = []
            for rep in range(3):
                t = time.perf_counter(); data = nx.prove([zz] * k, 30 + k); ts.append((time.perf_counter() - t) * 1e3)
            return float(np.median(ts)), data
        with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx:
            ms1, want1 = timed(nx, 1); ms8, want8 = timed(nx, 8)
        out['single_device'] = {'prove_ms': ms1, 'instances_8_ms': ms8}
        host = ck.bases.download()
        for G in shard_counts:
            t0 = time.perf_counter(); sb = aleo_amd.ShardedBases(host, devices=[0] * G, precompute=True); setup_s = time.perf_counter() - t0
            try:
                ck.bases.attach_shards(sb, 1 << 16)
                with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx:
                    a1, d1 = timed(nx, 1); a8, d8 = timed(nx, 8)
                if d1 != want1 or d8 != want8: raise SystemExit('bench: the proof against the sharded key differs from the single-device proof')
                out['shards_%d' % G] = {'prove_ms': a1, 'instances_8_ms': a8, 'constraints_per_s_8': 8 * n / a8 * 1e3, 'pin_and_tables_s': setup_s, 'bytes_equal_single_device': True}
                if G == 2:                                   # the same with every transform of >= 2^16 elements over the shards too (aleo_mi355x_ntt_fr_sharded_device's path: slabs by peer copy)
                    ck.bases.attach_shards(sb, 1 << 16, transforms_from=1 << 16)
                    with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx:
                        t1, e1 = timed(nx, 1)
                    if e1 != want1: raise SystemExit('bench: the proof with sharded transforms differs from the single-device proof')
                    out['shards_2_with_transforms'] = {'prove_ms': t1, 'bytes_equal_single_device': True,
                                                       'what': 'commitments AND transforms of >= 2^16 elements over the shards (default: transforms from 2^24 elements); one card: pure overhead of the split'}
            finally:
                ck.bases.attach_shards(None); sb.close()
        del host
        return out
    finally:
        ck.close()


def varuna_prove_big(synth, lg):
    """The same prover on a circuit the size of the headline MSM: key synthesis, one proof, eight instances in one proof (native entry points) — on the
    uniform synthetic circuit, and on the bit-heavy one with the Lagrange-basis powers pinned (commit_lagrange in the first round)."""
    out = _varuna_big_once(synth, lg, False, False)
    try: out['bit_heavy_witness_commit_lagrange'] = _varuna_big_once(synth, lg, True, True)
    except Exception as e: out['bit_heavy_witness_commit_lagrange'] = {'error': repr(e)[:300]}
    try: out['sharded_rehearsal'] = varuna_sharded_rehearsal(synth, lg)
    except SystemExit: raise
    except Exception as e: out['sharded_rehearsal'] = {'error': repr(e)[:300]}
    return out


def varuna_cpu(synth, lg):
    """cpu_baseline leg: the device's proof for a 2^lg circuit checked by the restatement's verifier, byte-compared with the restatement's own
    proof, and the restatement prover (plain Python integers, one core) timed on the same instance."""
    from aleo_amd import varuna
    from oracle import varuna_ref as V
    n, csr, z, zz, ck, D = _varuna_instance(synth, lg, 90 + lg)
    try:
        ix = varuna.CircuitIndex(csr, n, 4, len(z) - 4, ck)
        data = varuna.prove_native(ix, zz, 4242)
        def rows(m):
            ptr, col, val = csr[m]
            return [[(int(col[k]), synth.limbs_to_int(val[k])) for k in range(ptr[i], ptr[i + 1])] for i in range(len(ptr) - 1)]
        c = V.Circuit(n, 4, len(z) - 4, rows('a'), rows('b'), rows('c'))
        setup = V.Setup(VARUNA_TAU, VARUNA_S, D); idx = V.Index(c, setup)
        rand = V.random_stream(4242, c.n_h)
        t0 = time.perf_counter(); _, want = V.prove(idx, setup, z, rand); dt = time.perf_counter() - t0
        return {'constraints': n, 'device_proof_verifies': bool(V.verify_pairing(idx, setup.verifier_key(c), z[:4], data)), 'verifier': 'pairing products over public G2 elements (oracle/pairing.py), no trapdoor', 'device_proof_equals_restatement': bool(data == want),
                'restatement_prove_s': dt, 'restatement_constraints_per_s': n / dt, 'cores': 1,
                'note': 'oracle/varuna_ref.py in plain Python integers: a checker, not a tuned CPU prover and not the Rust binary'}
    finally:
        ck.close()


def proxy_schedule(lg):
    """[(op, arg, size)] of one Varuna prove_batch for one circuit / one instance with 2^lg constraints, variables and
    non-zeros per matrix (snarkvm-algorithms 0.14.5 snark/varuna/ahp/prover/round_functions [UPSTREAM-RECALL]):
    12 KZG MSMs in five groups (3 + 1, 2, 3, 1 commitments, 2 openings), the 23 transforms between them, and the two
    matrix-vector products z_a = A z, z_b = B z that precede round 1.
    ('commit', [(kind, size), ...]) = the commitments of one round; ('open', [size, ...]) = witness polynomials + their commitments;
    ('ntt', (direction, coset), size, count) = `count` independent transforms of one size issued as ONE batched call."""
    H = 1 << lg
    ops = []
    ops += [('spmv', 3, H)] * 2                                                         # before round 1: z_a = A z, z_b = B z (3 non-zeros per constraint row)
    ops += [('ntt', (1, 0), H, 3)]                                                      # round 1: interpolate w, z_a, z_b (independent: one batched call)
    ops += [('commit', [('witness', H)] * 3 + [('uniform', 3 * H)])]                    #          w, z_a, z_b, mask_poly (degree 3|H|)
    ops += [('ntt', (0, 1), 4 * H, 4)] + [('vec', 0, 4 * H)] * 6 + [('ntt', (1, 1), 4 * H, 1)]      # round 2: h_1 on the 4|H| coset
    ops += [('ntt', (0, 0), H, 2), ('commit', [('uniform', H), ('uniform', 2 * H)])]                #          g_1, h_1
    ops += [('ntt', (0, 0), H, 6)] + [('inv', 0, H)] * 3 + [('vec', 0, H)] * 12 + [('ntt', (1, 0), H, 3)]   # round 3: g_a, g_b, g_c over K (three matrices side by side)
    ops += [('commit', [('uniform', H)] * 3)]
    ops += [('ntt', (0, 1), 2 * H, 3)] + [('vec', 0, 2 * H)] * 4 + [('ntt', (1, 1), 2 * H, 1), ('commit', [('uniform', H)])]   # round 4: h_2
    ops += [('open', [3 * H, 3 * H])]                                                   # two batched KZG opening proofs
    return ops


def _proxy_csr(rows, per_row):
    """A synthetic R1CS-matrix shape: `per_row` non-zeros in every row, columns spread by a fixed stride (CSR with u32 indices)."""
    row_ptr = (np.arange(rows + 1, dtype=np.uint64) * per_row).astype(np.uint32)
    k = np.arange(rows * per_row, dtype=np.uint64)
    col = ((k * np.uint64(2654435761)) % np.uint64(rows)).astype(np.uint32)
    return row_ptr, col


def proof_proxy_gpu(aleo_amd, synth, torch, dev, lg, reps=3):
    from aleo_amd import poly, wire
    H = 1 << lg
    pb = aleo_amd.PinnedBases.generate_multiples(synth.generator_affine104(), 1, 3 * H).precompute()
    buf = torch.from_numpy(synth.uniform_scalars(16 * H, 0xA1E00010 + lg).view(np.int64)).to(dev)     # canonical < r, read as Montgomery; room for 4 x 4H side by side
    aux = buf.clone()
    wit = [torch.from_numpy(wire.fr_from_bytes(synth.witness_like_scalars(H, 0xA1E00012 + lg + 64 * j).view(np.uint8).reshape(-1, 32)).view(np.int64)).to(dev) for j in range(3)]
    quo = [torch.empty((3 * H, 4), dtype=torch.int64, device=dev) for _ in range(2)]
    z = synth.uniform_scalars(2, 0xA1E00013 + lg)
    csr = _proxy_csr(H, 3)
    d_rp = torch.from_numpy(csr[0].view(np.int32)).to(dev); d_ci = torch.from_numpy(csr[1].view(np.int32)).to(dev)
    d_vals = torch.from_numpy(synth.uniform_scalars(csr[1].shape[0], 0xA1E00014 + lg).view(np.int64)).to(dev)
    doms = {}
    ops = proxy_schedule(lg)
    torch.cuda.synchronize()

    def run():
        t_msm = 0.0
        for op, arg, size, cnt in [(o[0], o[1], o[2] if len(o) > 2 else None, o[3] if len(o) > 3 else 1) for o in ops]:
            if op in ('commit', 'open'):
                torch.cuda.synchronize()             # the queued transforms finish first, so the split below is honest
                t0 = time.perf_counter()
                if op == 'commit':
                    ptrs, lens, w = [], [], 0
                    for kind, m in arg:
                        if kind == 'witness': ptrs.append(wit[w].data_ptr()); w += 1
                        else: ptrs.append(buf.data_ptr() + 32 * (len(ptrs) * 64))       # distinct (overlapping) windows of the resident vector
                        lens.append(m)
                    aleo_amd.KZG10.commit_batch_device(pb, ptrs, lens)                   # Montgomery -> canonical on the device, shared launches
                else:
                    for j, m in enumerate(arg):
                        poly.divide_by_linear_device(quo[j].data_ptr(), 0, buf.data_ptr() + 32 * 64 * j, m, z[j])
                    torch.cuda.synchronize()
                    aleo_amd.KZG10.commit_batch_device(pb, [q.data_ptr() for q in quo], [m - 1 for m in arg])
                t_msm += time.perf_counter() - t0
            elif op == 'ntt':
                d = doms.get(size) or doms.setdefault(size, aleo_amd.EvaluationDomain(size))
                d.ntt_batch_device(buf.data_ptr(), cnt, 0, arg[0], arg[1], 1)            # hipStreamLegacy: enqueue only, ordered with torch's stream
            elif op == 'spmv':
                poly.spmv_device(aux.data_ptr(), d_rp.data_ptr(), d_ci.data_ptr(), d_vals.data_ptr(), buf.data_ptr(), size, 1)
            elif op == 'vec':
                poly.fr_vec_op_device(aux.data_ptr(), aux.data_ptr(), buf.data_ptr(), size, 0, 1)
            else:
                poly.batch_inversion_device(aux.data_ptr(), size, 1)
        torch.cuda.synchronize()
        return t_msm
    run()
    ts, tm = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); m = run(); ts.append(time.perf_counter() - t0); tm.append(m)
    pb.close()
    dt = float(np.median(ts)); m = float(np.median(tm))
    n_msm = sum(len(o[1]) for o in ops if o[0] in ('commit', 'open'))
    return {'constraints': H, 'ms': dt * 1e3, 'constraints_per_s': H / dt, 'msm_ms': m * 1e3, 'ntt_and_field_ms': (dt - m) * 1e3,
            'n_msm': n_msm, 'n_msm_calls': sum(1 for o in ops if o[0] in ('commit', 'open')), 'n_ntt': sum(o[3] for o in ops if o[0] == 'ntt'),
            'n_ntt_calls': sum(1 for o in ops if o[0] == 'ntt')}


INDEX_NOTE = ('operator-level proxy for key synthesis (Process::synthesize_key / vm.deploy -> Varuna circuit indexing, '
              '/root/reference/wasm/src/programs/manager/mod.rs:164-177, rust/src/program/deploy.rs:142,151): per R1CS matrix A, B, C the four '
              'arithmetisation polynomials row, col, row_col, row_col_val over the non-zero domain K (batch inversion + products on evaluations), '
              'interpolated (12 iNTTs in ONE batched call) and committed (12 commitments in ONE batched call) [schedule UPSTREAM-RECALL]; not a key')


def index_proxy_gpu(aleo_amd, synth, torch, dev, lg, reps=3):
    """Key-synthesis shape at |K| = 2^lg non-zeros per matrix."""
    from aleo_amd import poly
    K = 1 << lg
    pb = aleo_amd.PinnedBases.generate_multiples(synth.generator_affine104(), 1, K).precompute()
    ev = torch.from_numpy(synth.uniform_scalars(12 * K, 0xA1E00040 + lg).view(np.int64)).to(dev)      # 12 evaluation vectors over K, contiguous
    aux = ev.clone()
    d = aleo_amd.EvaluationDomain(K)
    torch.cuda.synchronize()

    def run():
        for m in range(3):                                   # per matrix: denominators inverted in one pass, then the products that form the four vectors
            poly.batch_inversion_device(aux.data_ptr() + 32 * 4 * m * K, K, 1)
            for j in range(4): poly.fr_vec_op_device(ev.data_ptr() + 32 * (4 * m + j) * K, ev.data_ptr() + 32 * (4 * m + j) * K, aux.data_ptr() + 32 * 4 * m * K, K, 0, 1)
        d.ntt_batch_device(ev.data_ptr(), 12, 0, 1, 0, 1)    # 12 iNTTs, one call
        torch.cuda.synchronize()
        return aleo_amd.KZG10.commit_batch_device(pb, [ev.data_ptr() + 32 * j * K for j in range(12)], [K] * 12)
    run(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
    pb.close()
    dt = float(np.median(ts))
    return {'nonzeros_per_matrix': K, 'ms': dt * 1e3, 'polynomials': 12, 'nonzeros_per_s': 3 * K / dt}


def proof_proxy_cpu(c, aleo_amd, synth, lg, cores):
    """The same schedule through the oracle on `cores` host threads (MSMs, transforms, element-wise field work)."""
    H = 1 << lg
    with aleo_amd.PinnedBases.generate_multiples(synth.generator_affine104(), 1, 3 * H) as pb:
        bases = pb.download()
    buf = synth.uniform_scalars(4 * H, 0xA1E00010 + lg); aux = buf.copy()
    s_uni = synth.uniform_scalars(3 * H, 0xA1E00011 + lg); s_wit = synth.witness_like_scalars(H, 0xA1E00012 + lg)
    z = synth.uniform_scalars(2, 0xA1E00013 + lg)
    t0 = time.perf_counter(); t_msm = 0.0
    for o in proxy_schedule(lg):
        op, arg = o[0], o[1]
        if op == 'commit':
            t1 = time.perf_counter()
            for kind, m in arg: c.msm_g1(bases[:m], (s_wit if kind == 'witness' else s_uni)[:m], threads=cores, variant=3)
            t_msm += time.perf_counter() - t1
        elif op == 'open':
            t1 = time.perf_counter()
            for j, m in enumerate(arg):
                q, _ = c.fr_divide_by_linear(buf[:m], z[j]); c.msm_g1(bases[:m - 1], q, threads=cores, variant=3)
            t_msm += time.perf_counter() - t1
        elif op == 'spmv':
            rp, ci = _proxy_csr(o[2], arg)
            aux[:o[2]] = c.fr_spmv_mt(rp, ci, s_uni[:ci.shape[0]], buf[:o[2]], cores)
        elif op == 'ntt':
            for _ in range(o[3]): buf[:o[2]] = c.ntt_fr(buf[:o[2]], 0, arg[0], arg[1], threads=cores)
        elif op == 'vec':
            aux[:o[2]] = c.fr_vec_op_mt(aux[:o[2]], buf[:o[2]], 0, cores)
        else:
            aux[:o[2]] = c.fr_batch_inverse_mt(aux[:o[2]], cores)
    dt = time.perf_counter() - t0
    return {'constraints': H, 'seconds': dt, 'constraints_per_s': H / dt, 'msm_seconds': t_msm, 'msm_threads': cores, 'other_threads': cores,
            'host_cpus': os.cpu_count(), 'usable_cpus': effective_cpus(),
            'what': 'restatement (oracle.c), not the Rust binary: MSMs with every window split over the threads, transforms with the butterflies of every stage split over the threads, '
                    'element-wise field work / batch inversion / matrix rows in per-thread chunks; the division by X - z stays serial'}


def effective_cpus():
    """Host cores this process may actually use: the cgroup CPU quota when there is one (a GPU box reports all 256 hardware threads
    but grants a share of them), else the scheduler affinity."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max': n = min(n, max(1, int(int(q) / int(per) + 0.5)))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()); per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0: n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def crossover(c, aleo_amd, bases, scalars, cores):
    """Where the drop-in pays (`if n >= MI355X_MIN_MSM` / `size >= MI355X_MIN_NTT` in the Rust stub): the ONE-SHOT entry points on host
    buffers — aleo_mi355x_msm_g1 (104-byte bases + scalars uploaded, nothing cached) and aleo_mi355x_ntt_fr (in place on host memory) — against the
    restatement on all usable host cores, 2^6 .. 2^20.  The first size from which the GPU call is faster for good is the threshold a maintainer sets
    (the library's defaults: aleo_mi355x_min_msm / aleo_mi355x_min_ntt, overridable by ALEO_MI355X_MIN_MSM / ALEO_MI355X_MIN_NTT)."""
    from aleo_amd import synth
    rows = []
    def best(fn, reps):
        fn(); ts = []
        for _ in range(reps):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        return float(np.median(ts)) * 1e3
    x_all = c.fr_to_mont(synth.uniform_scalars(1 << 20, 0xA1E00090))
    for lg in range(6, 21, 2):
        n = 1 << lg; B = np.ascontiguousarray(bases[:n]); S = np.ascontiguousarray(scalars[:n]); reps = 5 if lg <= 16 else 3
        g_msm = best(lambda: aleo_amd.VariableBase.msm(B, S), reps)
        c_msm = best(lambda: c.msm_g1(B, S, threads=cores, variant=3), reps)
        c_msm1 = best(lambda: c.msm_g1(B, S, threads=1, variant=1), 2) if lg <= 14 else None
        d = aleo_amd.EvaluationDomain(n); x = x_all[:n].copy()
        g_ntt = best(lambda: d.fft_in_place(x), reps)
        c_ntt = best(lambda: c.ntt_fr(x, 0, 0, 0, threads=cores), reps)
        c_ntt1 = best(lambda: c.ntt_fr(x, 0, 0, 0), reps)
        rows.append({'lg_n': lg, 'msm_gpu_one_shot_ms': g_msm, 'msm_cpu_ms': c_msm if c_msm1 is None else min(c_msm, c_msm1), 'ntt_gpu_host_buffer_ms': g_ntt, 'ntt_cpu_ms': min(c_ntt, c_ntt1)})
    def first_win(gk, ck):
        lg_win = None
        for r in reversed(rows):
            if r[gk] < r[ck]: lg_win = r['lg_n']
            else: break
        return lg_win
    return {'rows': rows, 'cpu_threads': cores, 'msm_gpu_wins_from_lg': first_win('msm_gpu_one_shot_ms', 'msm_cpu_ms'), 'ntt_gpu_wins_from_lg': first_win('ntt_gpu_host_buffer_ms', 'ntt_cpu_ms'),
            'library_defaults': {'min_msm': int(aleo_amd.lib().aleo_mi355x_min_msm()), 'min_ntt': int(aleo_amd.lib().aleo_mi355x_min_ntt())},
            'note': 'CPU = the C restatement (best of one thread and all usable threads), not the Rust binary: snarkVM on the same cores may be faster or slower by a small factor; '
                    'the GPU side is the cold two-line drop-in (no pinned SRS, no table) — with bases_pin the MSM crossover moves down to the launch-latency floor'}


def synth_mod():
    from aleo_amd import synth
    return synth


def cpu_baseline(args, pb, scalars, aleo_amd):
    """The oracle (C restatement of snarkVM's batched Pippenger, all host cores) on a bounded sample of the same
    workload; also re-checks GPU == oracle on that sample.  kind 'port': it is not the Rust binary."""
    from oracle import coracle as c
    ns = min(1 << args.cpu_sample_lg, len(scalars))
    ns = 1 << (ns.bit_length() - 1)
    # the restated algorithm (like snarkVM's rayon version) runs one window per thread: c = ln(n)+2 bits -> ceil(253/c)
    # windows is the most threads it can use, whatever the box has
    lg = ns.bit_length() - 1; nwin = -(-253 // (lg * 69 // 100 + 2))
    cores = min(effective_cpus(), nwin)
    bases = pb.download(0, ns)
    s = np.ascontiguousarray(scalars[:ns])
    c.msm_g1(bases[:256], s[:256], threads=1, variant=1)                    # load the library outside the timing
    t0 = time.perf_counter(); ref = c.msm_g1(bases, s, threads=cores, variant=1); dt_ref = time.perf_counter() - t0
    # ... and with the points of every window also split over the threads, so that the baseline uses the whole box
    all_cores = effective_cpus()
    t0 = time.perf_counter(); ref2 = c.msm_g1(bases, s, threads=all_cores, variant=3); dt = time.perf_counter() - t0
    if c.jac_to_int_point(ref2) != c.jac_to_int_point(ref): raise SystemExit('bench: the two CPU baselines disagree')
    got = aleo_amd.VariableBase.msm(pb, s)
    n1 = min(ns, 1 << 17)                                                   # "on all host cores and on 1 core"
    t0 = time.perf_counter(); c.msm_g1(bases[:n1], s[:n1], threads=1, variant=1); dt1 = time.perf_counter() - t0
    proxy = None
    if args.proof_proxy_lg and args.proof_proxy_cpu_lg:
        from aleo_amd import synth
        proxy = proof_proxy_cpu(c, aleo_amd, synth, args.proof_proxy_cpu_lg, effective_cpus())
    vcpu = varuna_cpu(synth_mod(), args.varuna_cpu_lg) if (args.varuna_lg and args.varuna_cpu_lg) else None
    if vcpu is not None and _BIG_PROOFS:
        # the large circuit's device proofs, checked by the restatement's verifier from the exported verifying key alone (no index, no trapdoor)
        from oracle import varuna_ref as V
        b = _BIG_PROOFS[0]; t0 = time.perf_counter()
        vk = V.VerifyingKey(b['vk'], 4); setup = V.Setup(VARUNA_TAU, VARUNA_S, b['max_degree']); keys = setup.verifier_key(vk.circuit)
        ok1 = bool(V.verify_pairing(vk, keys, b['public'], b['proof'])); ok8 = bool(V.verify(vk, setup, [b['public']] * 8, b['proof_8']))
        bad = bytearray(b['proof']); bad[400] ^= 1
        vcpu['large_circuit'] = {'constraints': b['constraints'], 'device_proof_verifies': ok1, 'eight_instance_proof_verifies': ok8,
                                 'tampered_proof_refused': not V.verify(vk, setup, b['public'], bytes(bad)), 'verify_s': time.perf_counter() - t0,
                                 'verifier': 'oracle/varuna_ref.py VerifyingKey: index commitments + domain sizes exported by the library; pairing products for the single proof'}
    cross = None
    if not args.no_crossover:
        try: cross = crossover(c, aleo_amd, bases, s, all_cores)
        except Exception as e: cross = {'error': repr(e)[:300]}
    return {'crossover': cross, 'proof_proxy': proxy, 'varuna': vcpu, 'one_core': {'value': n1 / dt1, 'unit': 'scalar-muls/s', 'cores': 1, 'seconds': dt1, 'sample': '2^%d-point prefix' % (n1.bit_length() - 1)},
            'value': ns / dt, 'unit': 'scalar-muls/s', 'cores': all_cores, 'kind': 'port', 'seconds': dt,
            'host_cpus': os.cpu_count(), 'usable_cpus': effective_cpus(),
            'window_parallel_only': {'value': ns / dt_ref, 'unit': 'scalar-muls/s', 'cores': cores, 'seconds': dt_ref,
                                     'note': 'the reference parallelises over windows only (rayon over ceil(253/c) windows): this is its shape'},
            'sample': '2^%d-point prefix of the same bases/scalars; C restatement of snarkvm-algorithms 0.14.5 batched MSM with the points of '
                      'every window also split over the threads (all host cores), not the Rust binary' % (ns.bit_length() - 1),
            'gpu_matches_oracle_on_sample': bool(c.jac_to_int_point(got) == c.jac_to_int_point(ref))}


if __name__ == '__main__':
    main()
